// bc7_exact.hpp -- integer-exact building blocks of the BC7 kernels (gfx950).
//
// BC7 texels and dequantised endpoints are integers in [0,255].  Large parts of what
// the reference computes in fp32 (kernel.ispc:763-803 moment sums, 1133-1193
// block_quant, 1198-1240 the sums of opt_endpoints) therefore never round: every
// intermediate is an integer below 2^24.  Those parts are restated here in integer
// arithmetic on the packed-math units (v_dot4_u32_u8, v_dot2_i32_i16, v_pk_mad_i16),
// which produce the SAME numbers with a third of the instructions.  Each function
// states the bound that makes it exact.  Whatever can round in the reference (PCA,
// endpoint quantisation, the tail of the least-squares solve) stays in fp32 with the
// pinned arithmetic of x86_math.hpp, operation for operation.
//
// Planar texel layout ("EO order") used by the dot4 sums: for each channel four
// dwords holding texels (0,2,4,6) (1,3,5,7) (8,10,12,14) (9,11,13,15), first texel in
// byte 0.  Nibble-packed index words expand to the same order with one AND (even
// texels) or shift+AND (odd texels); sums over texels are order-free because exact.
#pragma once
#include "bcn_core.hpp"

namespace itw {

#define BCN_TABLE_QUAL __device__ const
#include "bc7_bytemasks.h"
#undef BCN_TABLE_QUAL

typedef short s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s16x2 as_s16x2(uint32_t v) { return __builtin_bit_cast(s16x2, v); }
__device__ __forceinline__ uint32_t as_u32(s16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) { return as_u32(as_s16x2(a) - as_s16x2(b)); }
__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) { return as_u32(as_s16x2(a) + as_s16x2(b)); }
__device__ __forceinline__ int32_t dot2(uint32_t a, uint32_t b, int32_t c)
{
    return __builtin_amdgcn_sdot2(as_s16x2(a), as_s16x2(b), c, false);
}
__device__ __forceinline__ uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
// 16-bit VOP2 forms (2-cycle issue on gfx950, tools/ubench); operands are the low halves, the result is zero-extended.
// hipcc does not select them from C++ shorts here (it widens to v_mul_lo_u32 / v_mad_u64_u32), hence the asm.
__device__ __forceinline__ uint32_t mul_lo_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_mul_lo_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t add_u16(uint32_t a, uint32_t b) { uint32_t d; asm("v_add_u16 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ uint32_t add32_u16(uint32_t a) { uint32_t d; asm("v_add_u16 %0, 32, %1" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t ashr6_i16(uint32_t a) { uint32_t d; asm("v_ashrrev_i16 %0, 6, %1" : "=v"(d) : "v"(a)); return d; }
__device__ __forceinline__ uint32_t pack16(int32_t lo, int32_t hi) { return (uint32_t)lo | ((uint32_t)hi << 16); }   // both in [0,65535]
__device__ __forceinline__ int32_t imed3(int32_t v, int32_t lo, int32_t hi) { return min(max(v, lo), hi); }

// ISPC rcp(n) for n = 0..16 under the pinned arithmetic (x86_math.hpp), as bit patterns.  Subset sizes are
// wave-uniform in table-order scans, so `covariance = moments - s*s*rcp(n)` reads a scalar constant instead of
// gathering the RCPPS seed.  tests/test_gpu_math.py checks the table against the device's own ispc_rcp.
__device__ const uint32_t RCP_OF_COUNT[17] = {
    0xffc00000u, 0x3f7fffffu, 0x3effffffu, 0x3eaaaaaau, 0x3e7fffffu, 0x3e4cccccu, 0x3e2aaaaau, 0x3e124924u, 0x3dffffffu,
    0x3de38e38u, 0x3dccccccu, 0x3dba2e8bu, 0x3daaaaaau, 0x3d9d89d8u, 0x3d924924u, 0x3d888888u, 0x3d7fffffu};
__device__ __forceinline__ float rcp_of_count(int n) { return __uint_as_float(RCP_OF_COUNT[n]); }

// ---- the lane's block -------------------------------------------------------------------------
struct Tex {
    uint32_t w[16];        // RGBA8 as loaded, texel k
    uint32_t pl[4][4];     // planar bytes, EO order: pl[channel][dword]
    __device__ __forceinline__ float get(int p, int k) const { return (float)((w[k] >> (8 * p)) & 255u); }
    // (c0 | c1 << 16) and (c2 | c3 << 16) of texel k as 16-bit pairs; ALPHA=false leaves the c3 half zero
    __device__ __forceinline__ uint32_t pair01(int k) const { return __builtin_amdgcn_perm(0u, w[k], 0x0c010c00u); }
    template <bool ALPHA>
    __device__ __forceinline__ uint32_t pair23(int k) const
    {
        return ALPHA ? __builtin_amdgcn_perm(0u, w[k], 0x0c030c02u) : ((w[k] >> 16) & 255u);
    }
    // 4x4 byte transposes of the loaded words into the planar form (8 v_perm per 4 texels)
    __device__ __forceinline__ void make_planar()
    {
        const int grp[4][4] = {{0, 2, 4, 6}, {1, 3, 5, 7}, {8, 10, 12, 14}, {9, 11, 13, 15}};
#pragma unroll
        for (int d = 0; d < 4; d++) {
            const uint32_t a = w[grp[d][0]], b = w[grp[d][1]], c = w[grp[d][2]], e = w[grp[d][3]];
            const uint32_t t0 = __builtin_amdgcn_perm(b, a, 0x05010400u);   // a.r b.r a.g b.g
            const uint32_t t1 = __builtin_amdgcn_perm(b, a, 0x07030602u);   // a.b b.b a.a b.a
            const uint32_t u0 = __builtin_amdgcn_perm(e, c, 0x05010400u);
            const uint32_t u1 = __builtin_amdgcn_perm(e, c, 0x07030602u);
            pl[0][d] = __builtin_amdgcn_perm(u0, t0, 0x05040100u);
            pl[1][d] = __builtin_amdgcn_perm(u0, t0, 0x07060302u);
            pl[2][d] = __builtin_amdgcn_perm(u1, t1, 0x05040100u);
            pl[3][d] = __builtin_amdgcn_perm(u1, t1, 0x07060302u);
        }
    }
    // Compiler fence for candidate loops: everything derived from the texels is loop invariant and LICM would
    // hoist hundreds of conversions / products out of the shape loop into registers or scratch.
    __device__ __forceinline__ void fence()
    {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("" : "+v"(w[k]));
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int d = 0; d < 4; d++) asm volatile("" : "+v"(pl[c][d]));
    }
};

// ---- subset masks -------------------------------------------------------------------------------
struct SubsetMask {
    uint32_t bits;      // texel k in subset: bit k
    uint32_t bm[4];     // the same as byte masks over the planar layout
    int32_t n;          // texel count
};

// subset j (0..2) of table shape `shape`; wave-uniform when `shape` is (scalar loads), per lane otherwise
__device__ __forceinline__ SubsetMask subset_of(int shape, int j)
{
    const uint32_t masks = BCN_SUBSET_MASKS[shape];
    const uint32_t m0 = masks & 0xffffu, m1 = masks >> 16;
    const uint32_t* t = BCN_BYTEMASK + shape * 8;
    SubsetMask s;
    if (j == 0)      { s.bits = m0; for (int d = 0; d < 4; d++) s.bm[d] = t[d]; }
    else if (j == 1) { s.bits = m1; for (int d = 0; d < 4; d++) s.bm[d] = t[4 + d]; }
    else             { s.bits = ~(m0 | m1) & 0xffffu; for (int d = 0; d < 4; d++) s.bm[d] = ~(t[d] | t[4 + d]); }
    s.n = __builtin_popcount(s.bits);
    return s;
}

__device__ __forceinline__ SubsetMask whole_block()
{
    return SubsetMask{0xffffu, {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, 16};
}

// ---- second moments of a subset (kernel.ispc:763-803) as integers ---------------------------------
// Exact: a product of two texel values is <= 65025 and a sum of 16 of them <= 1 040 400 < 2^24, so the
// reference's float accumulators hold exactly these integers whatever the summation order.
template <int CH>
struct IStats {
    int32_t m[10];   // 00 01 02 03 11 12 13 22 23 33
    int32_t s[4];
    int32_t n;
};

template <int CH>
__device__ __forceinline__ void stats_int(IStats<CH>& st, const uint32_t (&pl)[4][4], const SubsetMask& sm)
{
    uint32_t r[4], g[4], b[4], a[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
        r[d] = pl[0][d] & sm.bm[d]; g[d] = pl[1][d] & sm.bm[d]; b[d] = pl[2][d] & sm.bm[d];
        a[d] = (CH == 4) ? (pl[3][d] & sm.bm[d]) : 0u;
    }
    uint32_t m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, s[4] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        m[0] = udot4(r[d], pl[0][d], m[0]); m[1] = udot4(r[d], pl[1][d], m[1]); m[2] = udot4(r[d], pl[2][d], m[2]);
        m[4] = udot4(g[d], pl[1][d], m[4]); m[5] = udot4(g[d], pl[2][d], m[5]);
        m[7] = udot4(b[d], pl[2][d], m[7]);
        s[0] = udot4(r[d], 0x01010101u, s[0]); s[1] = udot4(g[d], 0x01010101u, s[1]); s[2] = udot4(b[d], 0x01010101u, s[2]);
        if (CH == 4) {
            m[3] = udot4(r[d], pl[3][d], m[3]); m[6] = udot4(g[d], pl[3][d], m[6]); m[8] = udot4(b[d], pl[3][d], m[8]);
            m[9] = udot4(a[d], pl[3][d], m[9]);
            s[3] = udot4(a[d], 0x01010101u, s[3]);
        }
    }
#pragma unroll
    for (int i = 0; i < 10; i++) st.m[i] = (int32_t)m[i];
#pragma unroll
    for (int i = 0; i < 4; i++) st.s[i] = (int32_t)s[i];
    st.n = sm.n;
}

template <int CH>
__device__ __forceinline__ void stats_sub(IStats<CH>& a, const IStats<CH>& b)      // a -= b (exact)
{
#pragma unroll
    for (int i = 0; i < 10; i++) a.m[i] -= b.m[i];
#pragma unroll
    for (int i = 0; i < 4; i++) a.s[i] -= b.s[i];
    a.n -= b.n;
}

// ---- an exact LOWER BOUND of every two-subset encoding of a block (round 4: the bounded mode order, bc7.hip) ----------------
// Whatever endpoints a mode 1 / 3 encoding of a shape ends with (first fit or any refinement), each subset decodes to ROUNDED points of
// one segment: level = floor(L + 1/2) per channel with L on the segment between the two integer endpoints (kernel.ispc:1164-1170).  So a
// texel's error is >= (dist(texel, line) - sqrt(3)/2)_+^2, a subset's error is >= (sqrt(R) - sqrt(3)/2 sqrt(n))_+^2 with R = the sum of
// squared distances of its texels to their best line = trace - largest eigenvalue of the subset's scatter matrix, and the shape's error is
// the sum over its subsets (plus sum (255 - alpha)^2 under an RGBA profile).  The largest eigenvalue is bounded from ABOVE by
// ||M^2||_F^(1/2) (M = n x scatter as exact integers, scaled by 1 / trace so that nothing overflows or underflows; ITW_BOUND_SQUARINGS=2:
// ||M^4||_F^(1/4), 3 % fewer blocks visited on noisy content for 15 % more arithmetic per shape: measured slower), and every float
// rounding below is covered by a margin of 1e-5 of the trace: the value returned never exceeds the true bound.  CPU restatement and its
// check against part_fast's error of every shape: oracle/bc7_bound.c, tests/test_bc7_bound.py.
#ifndef ITW_BOUND_SQUARINGS
#define ITW_BOUND_SQUARINGS 1
#endif
__device__ const float BOUND_SLACK3[17] = {     // sqrt(3)/2 sqrt(n), rounded up
    0.f, 0.866026282f, 1.22474611f, 1.50000155f, 1.73205256f, 1.93649364f, 2.12132239f, 2.29129004f, 2.44949222f, 2.59807873f,
    2.73861551f, 2.87228417f, 3.0000031f, 3.12250209f, 3.24037361f, 3.35410523f, 3.46410513f};

// n x (residual of the subset about its best line), from below; >= 0
__device__ __forceinline__ float subset_residual_bound(const IStats<3>& st)
{
    const int32_t n = st.n;
    // n x scatter matrix: n sum(xy) - sum(x) sum(y) <= 16 x 1 040 400 < 2^24: exact integers, exact as floats
    const int32_t c00 = n * st.m[0] - st.s[0] * st.s[0], c01 = n * st.m[1] - st.s[0] * st.s[1], c02 = n * st.m[2] - st.s[0] * st.s[2];
    const int32_t c11 = n * st.m[4] - st.s[1] * st.s[1], c12 = n * st.m[5] - st.s[1] * st.s[2], c22 = n * st.m[7] - st.s[2] * st.s[2];
    const float t = (float)(c00 + c11 + c22);
    const float inv = __builtin_amdgcn_rcpf(fmaxf(t, 1.0f));
    const float a = (float)c00 * inv, b = (float)c01 * inv, c = (float)c02 * inv, d = (float)c11 * inv, e = (float)c12 * inv, f = (float)c22 * inv;
    // M^2, then M^4 (symmetric: six entries each)
    const float bb = b * b, cc = c * c, ee = e * e;
    const float A = a * a + bb + cc, B = a * b + b * d + c * e, C = a * c + b * e + c * f;
    const float D = bb + d * d + ee, E = b * c + d * e + e * f, F = cc + ee + f * f;
#if ITW_BOUND_SQUARINGS == 1
    const float off1 = B * B + C * C + E * E;
    const float fro1 = (A * A + D * D + F * F) + (off1 + off1);
    const float lam = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(fro1));                             // ||M^2||_F^(1/2)
#else
    const float BB = B * B, CC = C * C, EE = E * E;
    const float A2 = A * A + BB + CC, B2 = A * B + B * D + C * E, C2 = A * C + B * E + C * F;
    const float D2 = BB + D * D + EE, E2 = B * C + D * E + E * F, F2 = CC + EE + F * F;
    const float off = B2 * B2 + C2 * C2 + E2 * E2;
    const float fro2 = (A2 * A2 + D2 * D2 + F2 * F2) + (off + off);
    const float lam = __builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(__builtin_amdgcn_sqrtf(fro2)));     // >= largest eigenvalue / trace
#endif
    const float r = ((a + d + f) - lam) - 1e-5f;
    return fmaxf(r, 0.0f) * t;
}

// lower bound of the error of any mode 1 / 3 encoding of two-subset shape `shape` (wave-uniform), without the opaque term.  It also bounds
// every mode 7 encoding of the shape: that error is the colour part's plus the alpha part's, and the colour part is again rounded points of
// one segment per subset (the projection of the four-channel segment), within sqrt(3)/2 of a line in the three colour channels.
__device__ __forceinline__ float two_subset_bound(int shape, const uint32_t (&pl)[4][4], const IStats<3>& full)
{
    const SubsetMask sm = subset_of(shape, 0);
    IStats<3> s0, s1 = full;
    stats_int<3>(s0, pl, sm);
    stats_sub<3>(s1, s0);
    // rcp_of_count(n) lies just below 1 / n; v_sqrt_f32 is good to 1 ulp: the factor below covers it
    const float d0 = __builtin_amdgcn_sqrtf(subset_residual_bound(s0) * rcp_of_count(s0.n)) * 0.999999f - BOUND_SLACK3[s0.n];
    const float d1 = __builtin_amdgcn_sqrtf(subset_residual_bound(s1) * rcp_of_count(s1.n)) * 0.999999f - BOUND_SLACK3[s1.n];
    const float e0 = fmaxf(d0, 0.0f), e1 = fmaxf(d1, 0.0f);
    return (e0 * e0 + e1 * e1) * 0.999999f;
}

template <int CH>
__device__ __forceinline__ void stats_float(Stats<CH>& f, const IStats<CH>& st)
{
#pragma unroll
    for (int i = 0; i < 10; i++) f.m[i] = (float)st.m[i];
#pragma unroll
    for (int i = 0; i < 4; i++) f.s[i] = (float)st.s[i];
    f.n = (float)st.n;
}

// ---- PCA line fit of a subset (kernel.ispc:825-905): fp32, pinned arithmetic --------------------------
// `rn` = ISPC rcp(texel count).  Endpoints clamped to [0,255]; slots p >= CH are left untouched.
template <int CH>
__device__ __forceinline__ void fit_line(float (&ep)[2][4], const Tex& tx, uint32_t mask, const IStats<CH>& ist, float rn, const SeedTables& T)
{
    Stats<CH> st;
    stats_float<CH>(st, ist);
    float cv[10];
    covariance_of<CH>(cv, st, rn);
    float dc[4];
#pragma unroll
    for (int p = 0; p < CH; p++) dc[p] = st.s[p] * rn;

    const float inv_var = 1.0f / 65536.0f;
#pragma unroll
    for (int i = 0; i < 10; i++) cv[i] *= inv_var;
    const float eps = 0.001f * 0.001f;
    cv[0] += eps; cv[4] += eps; cv[7] += eps; cv[9] += eps;

    float axis[4];
    principal_axis<CH, 8, true>(axis, cv, T);

    // Extreme projections.  minps/maxps differ from v_min/max_f32 only when a NaN is involved (and in the sign of a
    // zero result, which the clamp below erases: (+-0)*axis + dc is dc, or +-0 -> max(.,0) = +0).  Texels and dc are
    // finite, so a NaN can only come from a non-finite axis (degenerate normalisation): those lanes take the
    // compare-and-select form, everyone else two 4-cycle instructions per texel instead of four plus wait states.
    float lo = __builtin_inff(), hi = -__builtin_inff();
    // The reference's `dot = 0; dot += ...` starts from +0: 0 + x == x except that it turns a -0 first product into +0,
    // and a zero of either sign in `dot` (hence in lo / hi) ends in the same endpoints: (+-0) * axis + dc is dc, or +0
    // when dc is +0 (dc, a sum of non-negative texels times a positive reciprocal, is never -0); hi - lo < 1 and the
    // +-0.5 adjustment do not see the sign either.  So the sum starts from its first product.
    float probe = axis[0];
#pragma unroll
    for (int p = 1; p < CH; p++) probe += axis[p];
    probe *= 0.0f;                                                              // NaN iff some axis component is NaN or inf
    if (__builtin_expect(probe != probe, 0)) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if ((mask >> k) & 1u) {
                float dot = axis[0] * (tx.get(0, k) - dc[0]);
#pragma unroll
                for (int p = 1; p < CH; p++) dot += axis[p] * (tx.get(p, k) - dc[p]);
                lo = fmin_x86(lo, dot);
                hi = fmax_x86(hi, dot);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if ((mask >> k) & 1u) {
                float dot = axis[0] * (tx.get(0, k) - dc[0]);
#pragma unroll
                for (int p = 1; p < CH; p++) dot += axis[p] * (tx.get(p, k) - dc[p]);
                lo = vmin_raw(lo, dot);                     // the instruction itself: fminf() adds a canonicalising v_max(x, x) per
                hi = vmax_raw(hi, dot);                     // operand the compiler cannot prove quiet (lo / hi cross basic blocks)
            }
        }
    }
    if (hi - lo < 1.0f) { lo -= 0.5f; hi += 0.5f; }
#pragma unroll
    for (int p = 0; p < CH; p++) {
        ep[0][p] = fclamp_num(lo * axis[p] + dc[p], 0.f, 255.f);
        ep[1][p] = fclamp_num(hi * axis[p] + dc[p], 0.f, 255.f);
    }
}

// ---- index selection (kernel.ispc:1133-1193 block_quant) in integers ----------------------------------
// One segment = the two dequantised endpoints of a subset.  Channels 0,1 travel as a 16-bit pair; with CH == 4
// channels 2,3 are a second pair, with CH == 3 channel 2 is a plain int32 (its arithmetic then runs on the
// 2-cycle 16/32-bit VOP2 forms instead of 4-cycle packed ops; measured issue costs: tools/ubench).
struct Segment {
    uint32_t a01, ba01;     // endpoint 0: c0 | c1 << 16;  endpoint 1 - endpoint 0 per channel (signed halves)
    uint32_t a23, ba23;     // CH == 4: the same for c2, c3.  CH == 3: c2 of endpoint 0 and (b2 - a2) as int32
    float k0, k1;           // BITS <= 3: -LEVELS/|b-a|^2 and 0.5 + 1/(4|b-a|^2);  BITS == 4: -|b-a|^2 and its reciprocal
};

// d[0], d[1]: endpoints as the decoder reconstructs them, [0,255].  CH==3 drops channel 3 from the metric.
template <int BITS, int CH>
__device__ __forceinline__ Segment make_segment(const int32_t (&d)[2][4])
{
    Segment s;
    s.a01 = pack16(d[0][0], d[0][1]);
    s.ba01 = pk_sub(pack16(d[1][0], d[1][1]), s.a01);
    int32_t dd;
    if (CH == 4) {
        s.a23 = pack16(d[0][2], d[0][3]);
        s.ba23 = pk_sub(pack16(d[1][2], d[1][3]), s.a23);
        dd = dot2(s.ba01, s.ba01, dot2(s.ba23, s.ba23, 0));
    } else {
        const int32_t ba2 = d[1][2] - d[0][2];
        s.a23 = (uint32_t)d[0][2];
        s.ba23 = (uint32_t)ba2;
        dd = dot2(s.ba01, s.ba01, ba2 * ba2);
    }
    const float dn = -(float)dd;
    const float r = (dd == 0) ? 0.0f : 1.0f / dn;                          // one IEEE divide per segment
    if (BITS <= 3) {
        s.k0 = (float)(1 << BITS) * r;                                      // = RN(LEVELS/dn): scaling by 2^k commutes with RN
        s.k1 = 0.5f - 0.25f * r;                                            // = RN(0.5 + RN(-0.25/dn))
    } else {
        s.k0 = dn;
        s.k1 = r;
    }
    return s;
}

// packed interpolation weight (w | w << 16) of index q; the format's tables (kernel.ispc:675-686)
template <int BITS>
__device__ __forceinline__ uint32_t weight_selector(int32_t q) { return (uint32_t)q * 0x00010001u + 0x0c000c00u; }
template <int BITS>
__device__ __forceinline__ uint32_t weight_from_selector(uint32_t sel)
{
    if (BITS == 2) return __builtin_amdgcn_perm(0u, 0x402b1500u, sel);                 // 0 21 43 64
    return __builtin_amdgcn_perm(0x40372e25u, 0x1b120900u, sel);                       // 0 9 18 27 37 46 55 64
}
__device__ __forceinline__ uint32_t weight_pair4(int32_t q) { return (((uint32_t)q * 68u + 8u) >> 4) * 0x00010001u; }   // 0 4 9 13 ... 60 64

// One texel against one segment: the reference projects the texel on the segment, rounds to an index q1 in
// [1, LEVELS-1], decodes q1-1 and q1 and keeps the closer (ties: q1).  Returns index and squared error.
// t01 = c0 | c1 << 16 of the texel; t23 = c2 | c3 << 16 (CH == 4) or c2 (CH == 3).
//
// Exactness.  With integer texel t and endpoints a, b:  N = sum (t-a)(b-a) and D = sum (b-a)^2 are integers
// below 2^18 (D <= 260100), so the reference's float N, D are exact; it forms p = RN(N/D) with a true divide
// (`proj /= div`, kernel.ispc:1158), x = RN(p*LEVELS + 0.5) (p*LEVELS is exact) and truncates, then clamps to
// [1, LEVELS-1].  Write y = N*LEVELS/D.
// (1) The reference's index equals clamp(floor(y + 0.5)).  For an integer m in [2, LEVELS-1]: if y + 0.5 >= m,
//     monotonicity of RN gives x >= m (m - 0.5 and m are representable).  Otherwise 2*N*LEVELS <= (2m-1)*D - 1, i.e.
//     y + 0.5 <= m - 1/(2D) with 1/(2D) >= 1.92e-6, while the two roundings raise x by at most 15.5*2^-24 (divide) plus
//     2^-21 (add, x < 16) = 1.4e-6 in total, so x < m.  Below 1 and above LEVELS-1 the clamp decides; truncation and
//     floor only differ for x in (-1,0), which clamps to 1 either way.
// (2) BITS <= 3: x~ = RN(M*k0 + k1) with M = -N exact, k0 = RN(-LEVELS/D), k1 = RN(0.5 + 1/(4D)).  By (1) the true value
//     y + 0.5 + 1/(4D) is >= m + 1/(4D) or <= m - 1/(4D).  |x~ - true| <= 6.5*2^-24 (k0) + 2^-25 (k1) + 2^-22 (fma, x~ < 8)
//     = 6.6e-7 for 3 bits -- used by 3-channel modes only, where D <= 195075 and 1/(4D) >= 1.28e-6 -- and
//     <= 2.5*2^-24 + 2^-25 + 2^-23 = 3.0e-7 for 2 bits, where 1/(4D) >= 9.6e-7 even with 4 channels.  Hence
//     floor(x~) = floor(y + 0.5) wherever the clamp does not decide.
// (3) BITS == 4 (mode 6): q0 = RN(M*rn), rn = RN(1/-D); rem = M - q0*(-D) is exact in fp32 (an integer multiple of
//     ulp(q0) below 2^19 ulps); RN(q0 + rem*rn) = RN(N/D) because a ratio of integers < 2^18 is never within 2^-19 ulp
//     of a rounding boundary; then one FMA gives x exactly as the reference computes it.
// D == 0 (segment is a point): the reference gets 0/0 = NaN -> cvttps2dq INT_MIN -> clamp 1; here x = 0.5 -> 0 -> 1.
// Decode (kernel.ispc:1172-1173): (int)(((64-w)*a + w*b + 32)/64), all exact integers, = a + ((w*(b-a)+32) >> 6)
// with an arithmetic shift.  Errors are sums of <= 4 squares of integers in [-255,255].
template <int BITS, int CH>
__device__ __forceinline__ void select_texel(int32_t& q_out, int32_t& e_out, const Segment& sg, uint32_t t01, uint32_t t23)
{
    constexpr int LEVELS = 1 << BITS;
    const uint32_t at01 = pk_sub(sg.a01, t01);
    uint32_t at23;
    int32_t m;
    if (CH == 4) {
        at23 = pk_sub(sg.a23, t23);
        m = dot2(at01, sg.ba01, dot2(at23, sg.ba23, 0));
    } else {
        at23 = sg.a23 - t23;                                   // int32, [-255,255]
        m = dot2(at01, sg.ba01, (int32_t)at23 * (int32_t)sg.ba23);
    }
    const float mf = (float)m;
    float x;
    if (BITS <= 3) {
        x = __builtin_fmaf(mf, sg.k0, sg.k1);
    } else {
        float q = mf * sg.k1;
        const float rem = __builtin_fmaf(-q, sg.k0, mf);
        q = __builtin_fmaf(rem, sg.k1, q);
        x = __builtin_fmaf(q, (float)LEVELS, 0.5f);
    }
    const int32_t q1 = imed3((int32_t)x, 1, LEVELS - 1);

    uint32_t w0, w1;
    if (BITS <= 3) {
        const uint32_t sel = weight_selector<BITS>(q1);
        w1 = weight_from_selector<BITS>(sel);
        w0 = weight_from_selector<BITS>(sel - 0x00010001u);
    } else {
        w1 = weight_pair4(q1); w0 = weight_pair4(q1 - 1);
    }
    const s16x2 k32 = {32, 32}, six = {6, 6};
    const uint32_t x0_01 = pk_add(as_u32((as_s16x2(w0) * as_s16x2(sg.ba01) + k32) >> six), at01);
    const uint32_t x1_01 = pk_add(as_u32((as_s16x2(w1) * as_s16x2(sg.ba01) + k32) >> six), at01);
    int32_t e0, e1;
    if (CH == 4) {
        const uint32_t x0_23 = pk_add(as_u32((as_s16x2(w0) * as_s16x2(sg.ba23) + k32) >> six), at23);
        const uint32_t x1_23 = pk_add(as_u32((as_s16x2(w1) * as_s16x2(sg.ba23) + k32) >> six), at23);
        e0 = dot2(x0_01, x0_01, dot2(x0_23, x0_23, 0));
        e1 = dot2(x1_01, x1_01, dot2(x1_23, x1_23, 0));
    } else {
        // third channel on the 2-cycle 16-bit VOP2 forms (results in bits 15:0, upper half zero on gfx9):
        // |w*(b-a) + 32| <= 16352 and |x| <= 255 fit int16; x*x <= 65025 fits uint16
        const uint32_t x0 = add_u16(ashr6_i16(add32_u16(mul_lo_u16(w0, sg.ba23))), at23);
        const uint32_t x1 = add_u16(ashr6_i16(add32_u16(mul_lo_u16(w1, sg.ba23))), at23);
        e0 = dot2(x0_01, x0_01, (int32_t)mul_lo_u16(x0, x0));
        e1 = dot2(x1_01, x1_01, (int32_t)mul_lo_u16(x1, x1));
    }
    const bool first = e0 < e1;
    q_out = first ? q1 - 1 : q1;
    e_out = min(e0, e1);
}

__device__ __forceinline__ Segment pick_segment(const Segment& s0, const Segment& s1, bool one)
{
    Segment r;
    r.a01 = one ? s1.a01 : s0.a01; r.a23 = one ? s1.a23 : s0.a23;
    r.ba01 = one ? s1.ba01 : s0.ba01; r.ba23 = one ? s1.ba23 : s0.ba23;
    r.k0 = one ? s1.k0 : s0.k0; r.k1 = one ? s1.k1 : s0.k1;
    return r;
}

// Whole block against up to three segments chosen per texel by `pattern` (2 bits per texel).  Used where the
// shape differs per lane (refinement of the lane's winner, ranked candidate lists) and for one-subset modes.
// Returns the block error; indices 4 bits per texel in qb.  The error sum is exact (< 2^24), as in the reference
// where `(int)err` per texel is summed in float.
template <int BITS, int CH, int PAIRS>
__device__ __forceinline__ int32_t select_block(uint32_t (&qb)[2], const Tex& tx, const Segment (&sg)[3], uint32_t pattern)
{
    int32_t total = 0;
    qb[0] = qb[1] = 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        Segment s = sg[0];
        if (PAIRS >= 2) {
            const uint32_t j = (pattern >> (2 * k)) & 3u;
            s = pick_segment(s, sg[1], j == 1u);
            if (PAIRS == 3) s = pick_segment(s, sg[2], j == 2u);
        }
        int32_t q, e;
        select_texel<BITS, CH>(q, e, s, tx.pair01(k), tx.template pair23<CH == 4>(k));
        if (k < 8) qb[0] |= (uint32_t)q << (4 * k); else qb[1] |= (uint32_t)q << (4 * (k - 8));
        total += e;
    }
    return total;
}

// ---- index selection through a per-segment palette in LDS ---------------------------------------------------
// The decoded colour of index q on a segment does not depend on the texel, so a table-order scan decodes every level
// ONCE per (subset, mode) into LDS and each texel then needs: the projection, one 16-byte LDS read of the two neighbouring
// levels, and per candidate one v_dot4_u32_u8 against the texel as loaded:  |P - t|^2 = |P|^2 - 2 P.t + |t|^2.  |t|^2 is
// the same for both candidates and sums to a per-block constant over the texels (each texel belongs to exactly one subset),
// which the caller adds once per shape: the block error is still the reference's exact integer.  Entry layout:
// {P0 | P1 << 8 | P2 << 16 | P3 << 24, -|P|^2}, level-major, lane-minor (`pal[level * PAL_STRIDE]`): any mix of levels
// across a wave is bank-conflict free.
//
// Projection (round 3).  N = sum (t-a)(b-a) = t.b - t.a - a.(b-a): the endpoints a, b as packed bytes ARE palette levels 0
// and LEVELS-1, so N is two v_dot4_u32_u8 against the texel word as loaded (the palette's alpha byte is 0 when CH == 3)
// and one subtraction -- no unpacking of the texel into 16-bit pairs.
// Index (round 3).  The reference's q1 equals clamp(floor(y + 0.5), 1, LEVELS-1), y = N*LEVELS/D (statement (1) above).
//   BITS == 2: q1 - 1 = [y + 0.5 >= 2] + [y + 0.5 >= 3] = [8N >= 3D] + [8N >= 5D] = [N >= ceil(3D/8)] + [N >= ceil(5D/8)]:
//     two integer thresholds per segment, two subtractions and two sign extractions per texel; no float, no divide.
//     D == 0 (the reference's 0/0 -> NaN -> INT_MIN -> clamp 1): thresholds that no N reaches.
//   BITS == 3: x' = fma(N, k0, k1') with k0 = RN(LEVELS/D) and k1' = RN(1/D)/4 (exact scaling) approximates y + 1/(4D) to
//     within 6.6e-7 (bound (2) above, now without the rounding of k1), and y + 1/(4D) is at least 1/(4D) >= 1.28e-6 away from
//     every half-integer m - 0.5 with m in [2, LEVELS-1] -- so x' is never a rounding tie there, and adding 1.5 * 2^23
//     (round to nearest at ulp 1) leaves floor(y + 0.5) in the low mantissa bits: one v_add_f32 instead of a conversion.
//     Outside [1.5, LEVELS-1.5) the clamp decides, and any rounding of x' lands on the clamped side.
struct PalSegment {
    uint32_t p0, p1;        // endpoints 0 / 1 as packed bytes (= palette levels 0 and LEVELS-1)
    int32_t nc;             // -sum a*(b-a): N = t.p1 - t.p0 + nc
    float k0, k1;           // BITS == 3: LEVELS/|b-a|^2 and 1/(4|b-a|^2)
    int32_t th1, th2;       // BITS == 2: ceil(3D/8) - 1, ceil(5D/8) - 1
};

constexpr int32_t PAL_MAGIC = 0x4b400000;                                   // bits of 1.5 * 2^23

template <int BITS, int CH, int PAL_STRIDE>
__device__ __forceinline__ PalSegment build_palette(uint2* pal, const int32_t (&d)[2][4])
{
    constexpr int LEVELS = 1 << BITS;
    PalSegment s;
    const uint32_t a01 = pack16(d[0][0], d[0][1]);
    const uint32_t ba01 = pk_sub(pack16(d[1][0], d[1][1]), a01);
    uint32_t a23, ba23;
    int32_t dd;
    if (CH == 4) {
        a23 = pack16(d[0][2], d[0][3]);
        ba23 = pk_sub(pack16(d[1][2], d[1][3]), a23);
        dd = dot2(ba01, ba01, dot2(ba23, ba23, 0));
        s.nc = -dot2(a01, ba01, dot2(a23, ba23, 0));
    } else {
        const int32_t ba2 = d[1][2] - d[0][2];
        a23 = (uint32_t)d[0][2];
        ba23 = (uint32_t)ba2;
        dd = dot2(ba01, ba01, ba2 * ba2);
        s.nc = -dot2(a01, ba01, d[0][2] * ba2);
    }
    if (BITS == 4) {                                                       // mode 6: the exactly rounded quotient of select_texel (statement (3))
        const float dn = -(float)dd;
        s.k0 = dn;
        s.k1 = (dd == 0) ? 0.0f : 1.0f / dn;
        s.th1 = s.th2 = 0;
    } else if (BITS == 3) {
        const float dn = -(float)dd;
        const float r = (dd == 0) ? 0.0f : 1.0f / dn;                      // = RN(-1/D): one IEEE divide per segment
        s.k0 = -((float)LEVELS * r);
        s.k1 = -0.25f * r;
        s.th1 = s.th2 = 0;
    } else {
        s.k0 = s.k1 = 0.f;
        s.th1 = (dd == 0) ? 0x3fffffff : ((3 * dd + 7) >> 3) - 1;         // stored minus one: [N >= th] = sign of (th - 1 - N)
        s.th2 = (dd == 0) ? 0x3fffffff : ((5 * dd + 7) >> 3) - 1;         // |N| < 2^19: 2^30 is unreachable and th - 1 - N cannot wrap
    }
    const s16x2 k32 = {32, 32}, six = {6, 6};
#pragma unroll
    for (int q = 0; q < LEVELS; q++) {
        constexpr int D = LEVELS - 1;
        const uint32_t w = (uint32_t)((q * 128 + D) / (2 * D)) * 0x00010001u;           // the format's weight, both halves
        // decoded c0 | c1 << 16; the end levels are the endpoints themselves ((0*d + 32) >> 6 = 0, (64*d + 32) >> 6 = d)
        const uint32_t x01 = (q == 0) ? a01 : (q == D) ? pk_add(ba01, a01)
                                            : pk_add(as_u32((as_s16x2(w) * as_s16x2(ba01) + k32) >> six), a01);
        uint32_t bytes;
        if (CH == 4) {
            const uint32_t x23 = (q == 0) ? a23 : (q == D) ? pk_add(ba23, a23)
                                                : pk_add(as_u32((as_s16x2(w) * as_s16x2(ba23) + k32) >> six), a23);
            bytes = __builtin_amdgcn_perm(x23, x01, 0x06040200u);                         // low byte of each half
        } else {
            const uint32_t x2 = (q == 0) ? a23 : (q == D) ? (uint32_t)d[1][2] : add_u16(ashr6_i16(add32_u16(mul_lo_u16(w, ba23))), a23);
            bytes = __builtin_amdgcn_perm(x2, x01, 0x0c040200u);
        }
        const uint32_t pp = udot4(bytes, bytes, 0u);                                      // |P|^2 from the packed bytes: one instruction
        pal[q * PAL_STRIDE] = make_uint2(bytes, 0u - pp);                                 // negated: 2 P.t - |P|^2 is one v_lshl_add
        if (q == 0) s.p0 = bytes;
        if (q == D) s.p1 = bytes;
    }
    return s;
}

// N of one texel (RGBA8 word as loaded) against a palette segment
__device__ __forceinline__ int32_t pal_project(const PalSegment& sg, uint32_t w)
{
    return (int32_t)(udot4(w, sg.p1, (uint32_t)sg.nc) - udot4(w, sg.p0, 0u));
}

// q1 - 1 in [0, LEVELS-2]: the lower of the two neighbouring levels the reference compares
template <int BITS>
__device__ __forceinline__ int32_t pal_lower_level(const PalSegment& sg, int32_t n)
{
    constexpr int LEVELS = 1 << BITS;
    if (BITS == 2) {
        // [n >= th] as the sign bit of (th - 1 - n), moved down by a logical shift: 0 or 1, never negative (the level indexes
        // LDS).  Written as instructions: from C++ the compiler turns the pattern back into v_cmp + v_cndmask + v_addc
        // (4-cycle forms with wait states between them); these are 2-cycle VOP2 instructions.
        uint32_t d1, d2, s1, s2;
        asm("v_sub_u32 %0, %1, %2" : "=v"(d1) : "v"(sg.th1), "v"(n));
        asm("v_sub_u32 %0, %1, %2" : "=v"(d2) : "v"(sg.th2), "v"(n));
        asm("v_lshrrev_b32 %0, 31, %1" : "=v"(s1) : "v"(d1));
        asm("v_lshrrev_b32 %0, 31, %1" : "=v"(s2) : "v"(d2));
        return (int32_t)(s1 + s2);
    }
    if (BITS == 4) {
        // 4-bit indices (mode 6): 1/(4D) is too small a margin for the biased FMA, so the quotient is formed exactly as in
        // select_texel -- q0 = RN(M*rn), one exact remainder, one correction = RN(N/D) -- with M = -N (an exact negation)
        const float mf = (float)(-n);
        float q = mf * sg.k1;
        const float rem = __builtin_fmaf(-q, sg.k0, mf);
        q = __builtin_fmaf(rem, sg.k1, q);
        const float x = __builtin_fmaf(q, (float)LEVELS, 0.5f);
        return imed3((int32_t)x, 1, LEVELS - 1) - 1;
    }
    // + (1.5 * 2^23 - 1): the low mantissa bits are floor(y + 0.5) - 1 = q1 - 1 before the clamp; PAL_MAGIC << 11 is 0 mod 2^32,
    // so the level's LDS offset is the clamped word shifted, no subtraction
    const float x = __builtin_fmaf((float)n, sg.k0, sg.k1) + 12582911.0f;
    return imed3(__float_as_int(x), PAL_MAGIC, PAL_MAGIC + LEVELS - 2) - PAL_MAGIC;
}

// One texel against a palette.  `w` = the texel as loaded.  e_out excludes |t|^2 (see above).
template <int BITS, int CH, int PAL_STRIDE>
__device__ __forceinline__ void select_texel_pal(int32_t& q_out, int32_t& e_out, const PalSegment& sg, const uint2* pal, uint32_t w)
{
    const int32_t q0 = pal_lower_level<BITS>(sg, pal_project(sg, w));
    const uint2* p = pal + q0 * PAL_STRIDE;                                 // one address, two reads a level apart
    const uint2 lo = p[0], hi = p[PAL_STRIDE];
    // f = 2 P.t - |P|^2 = -(|P - t|^2 - |t|^2): the smaller error is the larger f; ties go to q1 like the reference's `<`
    const int32_t f0 = (int32_t)((udot4(lo.x, w, 0u) << 1) + lo.y), f1 = (int32_t)((udot4(hi.x, w, 0u) << 1) + hi.y);
    const bool first = f0 > f1;
    q_out = first ? q0 : q0 + 1;
    e_out = -max(f0, f1);
}

// The scans only need a candidate's ERROR: which of the two neighbouring levels won, and the packed indices, matter for
// the winner alone, and the finish kernels recompute them from the winner's endpoints (one selection pass per mode and
// block instead of compare + select + shift-or per texel, mode and shape).
template <int BITS, int CH, int PAL_STRIDE>
__device__ __forceinline__ int32_t texel_error_pal(const PalSegment& sg, const uint2* pal, uint32_t w)
{
    const int32_t q0 = pal_lower_level<BITS>(sg, pal_project(sg, w));
    const uint2* p = pal + q0 * PAL_STRIDE;
    const uint2 lo = p[0], hi = p[PAL_STRIDE];
    const uint32_t d0 = udot4(lo.x, w, 0u), d1 = udot4(hi.x, w, 0u);
    const int32_t f0 = (int32_t)((d0 << 1) + lo.y), f1 = (int32_t)((d1 << 1) + hi.y);
    return max(f0, f1);                                                     // = -(error - |t|^2) of the better level
}

// Error of the texels of one subset (wave-uniform mask), WITHOUT the |t|^2 terms; `total` accumulates.
template <int BITS, int CH, int PAL_STRIDE>
__device__ __forceinline__ void subset_error_pal(int32_t& total, const Tex& tx, const PalSegment& sg, const uint2* pal, uint32_t mask)
{
#pragma unroll
    for (int k = 0; k < 16; k++)
        if ((mask >> k) & 1u)
            total -= texel_error_pal<BITS, CH, PAL_STRIDE>(sg, pal, tx.w[k]);
}

// Two palettes (the two modes of a family) against one texel, staged so that the dot products sit side by side: a
// texel is its own basic block here (scalar branches on the subset mask), so the only instruction-level parallelism the
// scheduler finds is what one texel offers, and a dot product followed at once by its consumer costs wait states.
template <int BITSA, int BITSB, int CH, int PAL_STRIDE>
__device__ __forceinline__ void texel_error2_pal(int32_t& ta, int32_t& tc, const PalSegment& sa, const uint2* pa,
                                                 const PalSegment& sc, const uint2* pc, uint32_t w)
{
    const int32_t na = pal_project(sa, w), nb = pal_project(sc, w);
    const int32_t qa = pal_lower_level<BITSA>(sa, na), qb = pal_lower_level<BITSB>(sc, nb);
    const uint2* p0 = pa + qa * PAL_STRIDE;
    const uint2* p1 = pc + qb * PAL_STRIDE;
    const uint2 la = p0[0], ha = p0[PAL_STRIDE], lb = p1[0], hb = p1[PAL_STRIDE];
    const uint32_t d0 = udot4(la.x, w, 0u), d1 = udot4(ha.x, w, 0u), d2 = udot4(lb.x, w, 0u), d3 = udot4(hb.x, w, 0u);
    const int32_t f0 = (int32_t)((d0 << 1) + la.y), f1 = (int32_t)((d1 << 1) + ha.y);
    const int32_t f2 = (int32_t)((d2 << 1) + lb.y), f3 = (int32_t)((d3 << 1) + hb.y);
    ta -= max(f0, f1);
    tc -= max(f2, f3);
}

template <int BITSA, int BITSB, int CH, int PAL_STRIDE>
__device__ __forceinline__ void subset_error2_pal(int32_t& ta, int32_t& tc, const Tex& tx, const PalSegment& sa, const uint2* pa,
                                                  const PalSegment& sc, const uint2* pc, uint32_t mask)
{
#pragma unroll
    for (int k = 0; k < 16; k++)
        if ((mask >> k) & 1u)
            texel_error2_pal<BITSA, BITSB, CH, PAL_STRIDE>(ta, tc, sa, pa, sc, pc, tx.w[k]);
}

// Texels of one subset (wave-uniform mask) against one or two palettes; accumulates errors WITHOUT the |t|^2 terms.
template <int BITS, int CH, int PAL_STRIDE>
__device__ __forceinline__ void select_subset_pal(uint32_t (&qb)[2], int32_t& total, const Tex& tx, const PalSegment& sg,
                                                  const uint2* pal, uint32_t mask)
{
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((mask >> k) & 1u) {
            int32_t q, e;
            select_texel_pal<BITS, CH, PAL_STRIDE>(q, e, sg, pal, tx.w[k]);
            if (k < 8) qb[0] |= (uint32_t)q << (4 * k); else qb[1] |= (uint32_t)q << (4 * (k - 8));
            total += e;
        }
    }
}

template <int BITSA, int BITSB, int CH, int PAL_STRIDE>
__device__ __forceinline__ void select_subset2_pal(uint32_t (&qa)[2], int32_t& ta, uint32_t (&qc)[2], int32_t& tc, const Tex& tx,
                                                   const PalSegment& sa, const uint2* pa, const PalSegment& sc, const uint2* pc, uint32_t mask)
{
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((mask >> k) & 1u) {
            int32_t q0, e0, q1, e1;
            select_texel_pal<BITSA, CH, PAL_STRIDE>(q0, e0, sa, pa, tx.w[k]);
            select_texel_pal<BITSB, CH, PAL_STRIDE>(q1, e1, sc, pc, tx.w[k]);
            if (k < 8) { qa[0] |= (uint32_t)q0 << (4 * k); qc[0] |= (uint32_t)q1 << (4 * k); }
            else       { qa[1] |= (uint32_t)q0 << (4 * (k - 8)); qc[1] |= (uint32_t)q1 << (4 * (k - 8)); }
            ta += e0; tc += e1;
        }
    }
}

// Whole block against one palette (one-subset modes); returns the exact block error (`tt` = sum of |texel|^2 over the block).
template <int BITS, int CH, int PAL_STRIDE>
__device__ __forceinline__ int32_t select_block_pal(uint32_t (&qb)[2], const Tex& tx, const PalSegment& sg, const uint2* pal, int32_t tt)
{
    int32_t total = tt;
    qb[0] = qb[1] = 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int32_t q, e;
        select_texel_pal<BITS, CH, PAL_STRIDE>(q, e, sg, pal, tx.w[k]);
        if (k < 8) qb[0] |= (uint32_t)q << (4 * k); else qb[1] |= (uint32_t)q << (4 * (k - 8));
        total += e;
    }
    return total;
}

// ---- palettes where the shape differs per lane (round 3) ---------------------------------------------------------------
// Refinement of a lane's own winner and the ranked candidate lists of the fast presets cannot branch on the subset mask (it is
// per lane), so they used the direct path (select_block: decode both neighbouring levels per texel with packed 16-bit math,
// ~150 issue cycles per texel incl. the per-texel choice of segment).  The palette path works there too: each of the shape's
// PAIRS subsets decodes its LEVELS colours once into its own stretch of the lane's LDS column (subset j at level offset
// j * LEVELS), and a texel picks its subset's projection constants (five selects per extra subset) and palette stretch by the
// pattern word -- ~75 cycles per texel, plus ~30 per decoded level.  Same projection, same index, same two candidate levels,
// same integer errors as select_block; `tt` = sum of |texel|^2 over the block (CH channels), the term the palette path leaves out.
__device__ __forceinline__ PalSegment pick_pal(const PalSegment& s0, const PalSegment& s1, bool one)
{
    PalSegment r;
    r.p0 = one ? s1.p0 : s0.p0; r.p1 = one ? s1.p1 : s0.p1; r.nc = one ? s1.nc : s0.nc;
    r.k0 = one ? s1.k0 : s0.k0; r.k1 = one ? s1.k1 : s0.k1;
    r.th1 = one ? s1.th1 : s0.th1; r.th2 = one ? s1.th2 : s0.th2;      // (the unused pair is dead code per BITS)
    return r;
}

template <int CH>
__device__ __forceinline__ int32_t block_norm2(const uint32_t (&pl)[4][4])
{
    uint32_t t = 0;
#pragma unroll
    for (int c = 0; c < CH; c++)
#pragma unroll
        for (int d = 0; d < 4; d++) t = udot4(pl[c][d], pl[c][d], t);
    return (int32_t)t;
}

// WANT_Q = false: the error alone (candidate scans)
template <int BITS, int CH, int PAIRS, int PAL_STRIDE, bool WANT_Q>
__device__ __forceinline__ int32_t select_block_lanes_pal(uint32_t (&qb)[2], const Tex& tx, const PalSegment (&sg)[3], const uint2* pal,
                                                          uint32_t pattern, int32_t tt)
{
    constexpr int LEVELS = 1 << BITS;
    int32_t total = tt;
    qb[0] = qb[1] = 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        PalSegment s = sg[0];
        const uint2* base = pal;
        if (PAIRS >= 2) {
            const uint32_t j = (pattern >> (2 * k)) & 3u;
            s = pick_pal(s, sg[1], j == 1u);
            if (PAIRS == 3) s = pick_pal(s, sg[2], j == 2u);
            base = pal + j * (uint32_t)(LEVELS * PAL_STRIDE);
        }
        if (WANT_Q) {
            int32_t q, e;
            select_texel_pal<BITS, CH, PAL_STRIDE>(q, e, s, base, tx.w[k]);
            if (k < 8) qb[0] |= (uint32_t)q << (4 * k); else qb[1] |= (uint32_t)q << (4 * (k - 8));
            total += e;
        } else {
            total -= texel_error_pal<BITS, CH, PAL_STRIDE>(s, base, tx.w[k]);
        }
    }
    return total;
}

// ---- least-squares endpoints for fixed indices (kernel.ispc:1198-1262 opt_endpoints) -------------------
// The sums are exact integers (sum q*t <= 16*15*255); the 2x2 solve is fp32 exactly as in the reference.
template <int BITS, int CH>
__device__ __forceinline__ void refit_line(float (&ep)[2][4], const uint32_t (&pl)[4][4], const uint32_t (&qb)[2], const SubsetMask& sm, const SeedTables& T)
{
    constexpr uint32_t LM1 = (1u << BITS) - 1u;
    constexpr float L1 = (float)LM1;
    const uint32_t qe[4] = {qb[0] & 0x0f0f0f0fu, (qb[0] >> 4) & 0x0f0f0f0fu, qb[1] & 0x0f0f0f0fu, (qb[1] >> 4) & 0x0f0f0f0fu};
    uint32_t sq_ = 0, sqq = 0, ssum[4] = {0, 0, 0, 0}, satb[4] = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < 4; d++) {
        const uint32_t qm = qe[d] & sm.bm[d];
        const uint32_t xm = (LM1 * 0x01010101u - qe[d]) & sm.bm[d];      // (L1 - q) per byte, no borrows: q <= L1
        const uint32_t one = sm.bm[d] & 0x01010101u;
        sq_ = udot4(qm, 0x01010101u, sq_);
        sqq = udot4(qm, qm, sqq);
#pragma unroll
        for (int p = 0; p < CH; p++) {
            ssum[p] = udot4(pl[p][d], one, ssum[p]);
            satb[p] = udot4(pl[p][d], xm, satb[p]);
        }
    }
    const float sum_q = (float)sq_, sum_qq = (float)sqq, cnt = (float)sm.n;
    const float cxx = cnt * (L1 * L1) - (2.0f * L1) * sum_q + sum_qq;
    const float cyy = sum_qq;
    const float cxy = L1 * sum_q - sum_qq;
    const float det = cxx * cyy - cxy * cxy;
    const float scale = L1 * ispc_rcp(det, T);
    const bool flat = fabsf(det) < 0.001f;
    const float rcnt = ispc_rcp(cnt, T);
#pragma unroll
    for (int p = 0; p < CH; p++) {
        const float sum = (float)ssum[p], atb1 = (float)satb[p];
        const float atb2 = L1 * sum - atb1;
        const float e0 = (atb1 * cyy - atb2 * cxy) * scale;
        const float e1 = (atb2 * cxx - atb1 * cxy) * scale;
        const float mean = sum * rcnt;
        ep[0][p] = flat ? mean : e0;
        ep[1][p] = flat ? mean : e1;
    }
}

} // namespace itw
