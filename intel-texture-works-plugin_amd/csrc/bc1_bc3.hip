// bc1_bc3.hip -- BC1 / BC3 encoder kernels for gfx950 (MI355X).
//
// Replaces kernel.ispc:231-614 (CompressBlocksBC1_ispc / CompressBlocksBC3_ispc)
// behind CompressBlocksBC1/BC3 (ispc_texcomp.cpp:417-425).
//
// Mapping: one 4x4 block per lane, consecutive lanes own consecutive blocks of a
// block row.  A block's texel row is 16 contiguous bytes, so each of the four
// row loads is one global_load_dwordx4 per lane and 1 KiB contiguous per wave --
// fully coalesced without an LDS transpose.  Outputs are 8 (BC1) or 16 (BC3)
// contiguous bytes per lane.  The kernel is a stream: 64 B in, 8/16 B out and
// ~1.1 k VALU instructions per block, so it sits near the HBM/VALU ridge and is bound
// by VALU issue; nothing matrix shaped for MFMA.  LDS holds only
// lookup tables (reciprocal seeds, 565 quantisation), staged once per workgroup.
//
// Arithmetic is the pinned x86 model of x86_math.hpp; float sums run serially in
// texel order k = 0..15 inside the lane, exactly like one ISPC program instance.
#include "x86_math.hpp"
#include "kernels.hpp"

// Register allocation target: four waves per SIMD (measured round 3: 5 waves = 96 VGPRs + spills is 16 % slower, 3 and 2 waves
// change nothing -- the kernel is bound by the issue cycles of its own instruction stream).  The A/B switches and
// probe builds that measured this live in tools/history/variants_r03_r04/bc1_bc3_r03_probes.hip.
constexpr int BC13_WAVES = 4;

namespace itw {

// ---- per-workgroup tables in LDS ------------------------------------------------------------------------------------
// The kernel is bound by VALU issue, not by HBM: every instruction that is not one of the reference's
// fp32 multiplies / adds is overhead worth removing.  Two table families, staged once per workgroup (the workgroup then
// walks several chunks of 256 blocks, so the staging is amortised):
//   * RCPPS / RSQRTPS seeds, packed to 16 bit as generated (x86_luts_packed.h); a lookup is one LDS read, one v_lshl_or to
//     rebuild the seed word and one subtraction of the operand's exponent field;
//   * 8 bit -> 5/6 bit endpoint quantisation (kernel.ispc:234-248: (t + (t >> 8)) >> 8 with t = v*31 + 128) and the
//     decoded value of the code (kernel.ispc:250-259: bit replication), as two byte tables each, indexed by the
//     truncated endpoint: packing an endpoint is a conversion and a byte read per channel.
// One ready-made image (9 KiB), built at compile time: a workgroup stages it with plain 16-byte copies.
namespace tables_src {
#define X86_LUT_QUAL static constexpr
#include "x86_luts_packed.h"
#undef X86_LUT_QUAL
}
struct Bc1Image {
    unsigned short rsq16[2048]; // RSQRTPS seeds, indexed by bits [23:13] of the operand (exponent LSB, top 10 mantissa bits)
    unsigned short rcp16[2048]; // RCPPS seeds, indexed by mantissa[22:12]
    uint8_t q[1024];            // q5 | q6 | d5 | d6
};
constexpr Bc1Image make_bc1_image()
{
    Bc1Image im{};
    for (int i = 0; i < 2048; i++) {
        im.rsq16[i] = tables_src::X86_RSQRT_SEED16[i ^ 0x400];
        im.rcp16[i] = tables_src::X86_RCP_SEED16[i];
    }
    for (int v = 0; v < 256; v++) {
        const int t5 = v * 31 + 128, t6 = v * 63 + 128;
        const int c5 = (t5 + (t5 >> 8)) >> 8, c6 = (t6 + (t6 >> 8)) >> 8;
        im.q[v] = (uint8_t)c5; im.q[256 + v] = (uint8_t)c6;
        im.q[512 + v] = (uint8_t)((c5 << 3) + (c5 >> 2)); im.q[768 + v] = (uint8_t)((c6 << 2) + (c6 >> 4));
    }
    return im;
}
__device__ const Bc1Image BC1_IMAGE = make_bc1_image();
static_assert(sizeof(Bc1Image) == 4096 + 4096 + 1024, "Bc1Image layout");

// Round 3: the image is 9 KiB (was 21): the seeds stay packed to 16 bit and a lookup expands its word with one v_lshl_or.
// At launch every workgroup of the chip copies the image at the same time and nothing overlaps that copy but the first
// texel loads: measured 2.4 us of a 29 us BC1 launch for 21 KiB x 2048 workgroups (tools/history/misc/gpu_probe_bc1.sh).
struct Bc1Tables {
    const unsigned short* rsq16;
    const unsigned short* rcp16;
    const uint8_t* q5;          // [256]  5-bit code of a byte value
    const uint8_t* q6;          // [256]  6-bit code
    const uint8_t* d5;          // [256]  value a decoder reconstructs from q5[v]
    const uint8_t* d6;          // [256]
};

constexpr int BC1_LDS_BYTES = (int)sizeof(Bc1Image);

__device__ __forceinline__ Bc1Tables stage_bc1_tables(unsigned char* lds, int tid, int nthreads)
{
    const uint4* src = reinterpret_cast<const uint4*>(&BC1_IMAGE);
    uint4* dst = reinterpret_cast<uint4*>(lds);
    for (int i = tid; i < BC1_LDS_BYTES / 16; i += nthreads) dst[i] = src[i];
    const Bc1Image* im = reinterpret_cast<const Bc1Image*>(lds);
    Bc1Tables B;
    B.rsq16 = im->rsq16; B.rcp16 = im->rcp16;
    B.q5 = im->q; B.q6 = im->q + 256; B.d5 = im->q + 512; B.d6 = im->q + 768;
    return B;
}

// ISPC rcp(v) = r * (2 - v*r) on the RCPPS seed r, for an operand known to be a positive ordinary number (exponent field
// 1..252: the seed's exponent is 253 - e, its mantissa the table's, i.e. seed = (0x7e800000 | s16 << 11) - (e << 23)), or +0
// where the caller shows the result does not matter: no range test, no divergent branch, no sign handling -- the general
// model (x86_math.hpp x86_rcpps: zero, denormal, huge, inf, NaN, negative operands) is never needed in this kernel, each
// call site says why.  For +0 the seed comes out as a finite 2^126-sized number instead of +inf.
__device__ __forceinline__ float rcp_nr_pos(float v, const Bc1Tables& B)
{
    const uint32_t x = __float_as_uint(v);
    const uint32_t t = 0x7e800000u | ((uint32_t)B.rcp16[(x >> 12) & 0x7ffu] << 11);
    const float r = __uint_as_float(t - (x & 0x7f800000u));
    float u = v * r;
    u = 2.0f - u;
    return r * u;
}
// ISPC rsqrt() of a positive ordinary operand (x86_rsqrtps_fast without its range test)
__device__ __forceinline__ float rsqrt_nr_pos(float v, const Bc1Tables& B)
{
    const uint32_t x = __float_as_uint(v);
    const uint32_t t = 0x5f000000u | ((uint32_t)B.rsq16[(x >> 13) & 0x7ffu] << 11);      // seed | 0.5's exponent, + 64 << 23
    const float is = __uint_as_float(t - (((x + 0x00800000u) >> 1) & 0x7f800000u));
    float a = v * is;
    a = a * is;
    a = 3.0f - a;
    a = is * a;
    return 0.5f * a;
}

template <int N> __device__ __forceinline__ float ubyte_f32(uint32_t w)
{
    float d;
    if (N == 0) asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(d) : "v"(w));
    if (N == 1) asm("v_cvt_f32_ubyte1 %0, %1" : "=v"(d) : "v"(w));
    if (N == 2) asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(d) : "v"(w));
    if (N == 3) asm("v_cvt_f32_ubyte3 %0, %1" : "=v"(d) : "v"(w));
    return d;
}
// v_min_f32 as an instruction, for operands known to be numbers (fminf() quiets operands it cannot prove quiet)
__device__ __forceinline__ float vmin_raw(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// ---- the lane's texels -------------------------------------------------------------------------------------------------
// Kept as aligned register pairs so that the sums whose products cannot round -- the covariance of the centred texels
// (multiples of 1/16 below 2^8: products are multiples of 1/256 below 2^16, exact in fp32) and the refit's index-weighted
// sums (small integers) -- can run on v_pk_fma_f32, two multiply-adds per issue slot: with an exact product
// fma(a, b, c) = round(c + a*b) is bit for bit the reference's separate multiply and add, in the same texel order.
typedef float f2 __attribute__((ext_vector_type(2)));
struct Texels {
    f2 rg[16];          // (R, G) of texel k
    f2 b2[8];           // B of texels (2j, 2j+1)
    __device__ __forceinline__ float r(int k) const { return rg[k].x; }
    __device__ __forceinline__ float g(int k) const { return rg[k].y; }
    __device__ __forceinline__ float b(int k) const { return (k & 1) ? b2[k >> 1].y : b2[k >> 1].x; }
    __device__ __forceinline__ float ch(int p, int k) const { return p == 0 ? r(k) : (p == 1 ? g(k) : b(k)); }
};
// c += a * b per half; SEL picks which halves of b feed the low / high result (v_pk_fma_f32 op_sel / op_sel_hi)
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); return c; }
__device__ __forceinline__ f2 pk_fma_alo_b(f2 a, f2 b, f2 c)      // (a.lo*b.lo, a.lo*b.hi)
{ asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(c) : "v"(a), "v"(b)); return c; }
__device__ __forceinline__ f2 pk_fma_a_bhi(f2 a, f2 b, f2 c)      // (a.lo*b.hi, a.hi*b.hi)
{ asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(a), "v"(b)); return c; }
__device__ __forceinline__ f2 pk_fma_a_blo(f2 a, f2 b, f2 c)      // (a.lo*b.lo, a.hi*b.lo)
{ asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "+v"(c) : "v"(a), "v"(b)); return c; }

// an endpoint channel already clamped to [0,255] (NaN clamps to 0): its truncation indexes the byte tables
__device__ __forceinline__ int32_t trunc_byte(float c) { return cvt_i32_sat(c); }

struct Endpoint { int32_t code; float dec[3]; };                                  // 565 code + the colour it decodes to

__device__ __forceinline__ Endpoint quantise565(float cr, float cg, float cb, const Bc1Tables& B)   // [kernel.ispc:234-259]
{
    const int32_t ir = trunc_byte(cr), ig = trunc_byte(cg), ib = trunc_byte(cb);
    Endpoint e;
    e.code = (int32_t)(((uint32_t)B.q5[ir] << 11) | ((uint32_t)B.q6[ig] << 5) | (uint32_t)B.q5[ib]);
    e.dec[0] = (float)B.d5[ir]; e.dec[1] = (float)B.d6[ig]; e.dec[2] = (float)B.d5[ib];
    return e;
}

// keep the four-colour mode: p0 >= p1                                            [kernel.ispc:517-520, 525-528]
__device__ __forceinline__ void order(Endpoint& a, Endpoint& b)
{
    const bool sw = a.code < b.code;
    const int32_t ca = sw ? b.code : a.code, cb = sw ? a.code : b.code;
    a.code = ca; b.code = cb;
#pragma unroll
    for (int p = 0; p < 3; p++) { const float x = sw ? b.dec[p] : a.dec[p], y = sw ? a.dec[p] : b.dec[p]; a.dec[p] = x; b.dec[p] = y; }
}

// Project the 16 texels on the endpoint segment and emit linear 2-bit indices.   [kernel.ispc:308-344]
// Also returns each texel's index as a float (the refit's weights) when WANT_Q.
// The index clamp(trunc(t), 0, 3) is taken in the float domain -- clamp to [0, 3] first (a NaN t clamps to 0 like the
// reference's INT_MIN; trunc and floor agree on [0, 3]) then v_floor -- so the refit's float weight is the same register, and
// the sixteen indices are packed by Horner steps acc*4 + q on two 8-texel halves (exact: below 2^16), converted once:
// per texel clamp + floor + mul + add instead of convert + clamp + shift + or (+ convert back for the refit).
template <bool WANT_Q>
__device__ __forceinline__ uint32_t project_indices(const Texels& px, const Endpoint& e0, const Endpoint& e1, const Bc1Tables& B,
                                                    f2 (&qf)[8])
{
    float dir[3];
    for (int p = 0; p < 3; p++) dir[p] = e1.dec[p] - e0.dec[p];

    float sq_norm = sq(dir[0]);                      // 0 + x*x = x*x exactly
    sq_norm += sq(dir[1]); sq_norm += sq(dir[2]);
    // sq_norm is an integer in [1, 3*255^2], or 0 when both endpoints decode alike.  The reference's rcp(0) is NaN (inf seed
    // through the Newton step), every projection is NaN and every index (int)NaN = INT_MIN clamps to 0.  Here rcp(+0) stays
    // finite (or overflows to inf after the * 3): dir = 0 * finite = 0 -> dot + bias = 0.5 -> index 0, or dir = 0 * inf =
    // NaN -> index 0 as in the reference.  Either way the same sixteen zeros.
    const float rs3 = rcp_nr_pos(sq_norm, B) * 3.0f;
    for (int p = 0; p < 3; p++) dir[p] *= rs3;

    float bias = 0.5f;
    for (int p = 0; p < 3; p++) bias -= e0.dec[p] * dir[p];

    // The products run two per instruction (v_pk_mul_f32: the same IEEE multiply per half) -- (R, G) of a texel against
    // (dir0, dir1), B of a texel pair against dir2 -- and the sums keep the reference's order: (R*d0 + G*d1) + B*d2, then + bias.
    const f2 d01 = {dir[0], dir[1]}, d22 = {dir[2], dir[2]};
    float half[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 7; j >= 0; j--) {                   // Horner: texel k ends up at bits 2k
        const f2 bb = px.b2[j] * d22;
#pragma unroll
        for (int t = 1; t >= 0; t--) {
            const int k = 2 * j + t;
            const f2 m = px.rg[k] * d01;
            float dot = m.x + m.y;                   // the reference's 0 + a = a: exact (the sign of a zero cannot reach q)
            dot += t ? bb.y : bb.x;
            const float q = __builtin_floorf(fclamp_num(dot + bias, 0.f, 3.f));
            if (WANT_Q) { if (t) qf[j].y = q; else qf[j].x = q; }
            float& h = half[k >> 3];
            h = h * 4.0f;
            h = h + q;
        }
    }
    return (uint32_t)half[0] | ((uint32_t)half[1] << 16);
}

// Least-squares endpoint update for fixed indices.               [kernel.ispc:419-480]
// Exact integer identities (every partial sum below 2^24): with S_p = sum q*px and A_p = 16*dc_p = sum px, the reference's
// atb1 = sum (3-q)*px = 3*A_p - S_p and atb2 = 3*A_p - atb1 = S_p; sum_q and sum_qq are bit counts of the index word
// (q = 2h + l: sum q = 2*pop(H) + pop(L), sum q*q = 4*pop(H) + pop(L) + 4*pop(H & L)).
__device__ __forceinline__ void refit_endpoints(float (&c0)[3], float (&c1)[3], const Texels& px, uint32_t bits, const f2 (&qf)[8],
                                                const float (&acc)[3], const float (&dc)[3], const Bc1Tables& B)
{
    if ((bits ^ (bits * 4u)) < 4u) {
        for (int p = 0; p < 3; p++) { c0[p] = dc[p]; c1[p] = dc[p]; }
        return;
    }
    // S = sum q*px: integers (<= 3*255*16), so packed multiply-adds are exact whatever the order
    f2 srg = {0.f, 0.f}, sb = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        srg = pk_fma_a_blo(px.rg[2 * j], qf[j], srg);            // (R, G) of texel 2j   times q(2j)
        srg = pk_fma_a_bhi(px.rg[2 * j + 1], qf[j], srg);        // (R, G) of texel 2j+1 times q(2j+1)
        sb = pk_fma(px.b2[j], qf[j], sb);                        // B of both texels, each times its q
    }
    const float s[3] = {srg.x, srg.y, sb.x + sb.y};
    const uint32_t lo = bits & 0x55555555u, hi = (bits >> 1) & 0x55555555u;
    const int32_t pl = __builtin_popcount(lo), ph = __builtin_popcount(hi), pb = __builtin_popcount(lo & hi);
    const float sum_q = (float)(2 * ph + pl), sum_qq = (float)(4 * ph + pl + 4 * pb);
    const float cxx = 144.0f - 6.0f * sum_q + sum_qq;
    const float cyy = sum_qq;
    const float cxy = 3.0f * sum_q - sum_qq;
    // the determinant is an exact integer (all factors <= 144) and, by Cauchy-Schwarz, zero only when all indices are equal --
    // the case that left above: a positive ordinary number here
    const float scale = 3.0f * rcp_nr_pos(cxx * cyy - cxy * cxy, B);
    for (int p = 0; p < 3; p++) {
        const float atb1 = 3.0f * acc[p] - s[p];
        const float atb2 = s[p];
        c0[p] = fclamp_num((atb1 * cyy - atb2 * cxy) * scale, 0.f, 255.f);
        c1[p] = fclamp_num((atb2 * cxx - atb1 * cxy) * scale, 0.f, 255.f);
    }
}

// Colour part: PCA axis by power iteration, endpoint pick, one refit pass.
// [kernel.ispc:494-533]
__device__ __forceinline__ void encode_color(const Texels& px, uint32_t out[2], const Bc1Tables& B)
{
    // channel sums: integers <= 4080, exact in any order -- packed adds over the (R, G) pairs and the B pairs
    float acc[3], dc[3];
    {
        f2 srg = px.rg[0], sbb = px.b2[0];
#pragma unroll
        for (int k = 1; k < 16; k++) srg = srg + px.rg[k];
#pragma unroll
        for (int j = 1; j < 8; j++) sbb = sbb + px.b2[j];
        acc[0] = srg.x; acc[1] = srg.y; acc[2] = sbb.x + sbb.y;
        for (int p = 0; p < 3; p++) dc[p] = acc[p] * 0.0625f;
    }

    // packed symmetric covariance  [rr rg rb gg gb bb]          [kernel.ispc:377-417]
    // centred texels are exact multiples of 1/16, their products exact: three packed multiply-adds per texel, texel order kept
    f2 c_rr_gg = {0.f, 0.f}, c_rg_rb = {0.f, 0.f}, c_gb_bb = {0.f, 0.f};
    const f2 dc_rg = {dc[0], dc[1]};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const f2 a = px.rg[k] - dc_rg;                           // (r, g)
        f2 gb;
        gb.x = a.y; gb.y = px.b(k) - dc[2];                     // (g, b)
        c_rr_gg = pk_fma(a, a, c_rr_gg);
        c_rg_rb = pk_fma_alo_b(a, gb, c_rg_rb);
        c_gb_bb = pk_fma_a_bhi(gb, gb, c_gb_bb);
    }
    float cv[6] = {c_rr_gg.x, c_rg_rb.x, c_rg_rb.y, c_rr_gg.y, c_gb_bb.x, c_gb_bb.y};
    cv[0] += 0.001f; cv[3] += 0.001f; cv[5] += 0.001f;

    // four power iterations from (1,1,1), renormalised after the 2nd and 4th  [kernel.ispc:184-205]
    float v[3] = {1.f, 1.f, 1.f};
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const float a0 = cv[0] * v[0] + cv[1] * v[1] + cv[2] * v[2];
        const float a1 = cv[1] * v[0] + cv[3] * v[1] + cv[4] * v[2];
        const float a2 = cv[2] * v[0] + cv[4] * v[1] + cv[5] * v[2];
        v[0] = a0; v[1] = a1; v[2] = a2;
        if (it & 1) {
            float n = 0.f;
            n += a0 * a0; n += a1 * a1; n += a2 * a2;
            const float rn = rsqrt_nr_pos(n, B);           // the diagonal carries +0.001: n is a positive ordinary number (see below)
            v[0] *= rn; v[1] *= rn; v[2] *= rn;
        }
    }

    // extreme projections -> endpoints                           [kernel.ispc:274-306]
    // The projections are finite (the diagonal carries +0.001, so the iteration never meets a zero or overflowing norm):
    // minps / maxps are ordinary minimum / maximum here; only the sign of a zero could differ and it is added to dc >= 0.
    // Two differences and two products per instruction (v_pk_add_f32 / v_pk_mul_f32), sums in the reference's order.
    float lo = 65536.0f, hi = 0.0f;
    const f2 v01 = {v[0], v[1]}, v22 = {v[2], v[2]}, dc22 = {dc[2], dc[2]};
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const f2 bb = (px.b2[j] - dc22) * v22;
#pragma unroll
        for (int t = 0; t < 2; t++) {
            const f2 m = (px.rg[2 * j + t] - dc_rg) * v01;
            float dot = m.x + m.y;
            dot += t ? bb.y : bb.x;
            lo = __builtin_fminf(lo, dot);
            hi = __builtin_fmaxf(hi, dot);
        }
    }
    if (hi - lo < 1.0f) { lo -= 0.5f; hi += 0.5f; }

    float nsq = v[0] * v[0];
    nsq += v[1] * v[1]; nsq += v[2] * v[2];
    const float rn = rcp_nr_pos(nsq, B);                 // |v|^2 of a just-normalised axis: about 1

    float c0[3], c1[3];
    for (int p = 0; p < 3; p++) {
        c0[p] = fclamp_num(dc[p] + lo * rn * v[p], 0.f, 255.f);
        c1[p] = fclamp_num(dc[p] + hi * rn * v[p], 0.f, 255.f);
    }

    Endpoint e0 = quantise565(c0[0], c0[1], c0[2], B), e1 = quantise565(c1[0], c1[1], c1[2], B);
    order(e0, e1);
    f2 qf[8];
    uint32_t idx = project_indices<true>(px, e0, e1, B, qf);

    refit_endpoints(c0, c1, px, idx, qf, acc, dc, B);
    e0 = quantise565(c0[0], c0[1], c0[2], B); e1 = quantise565(c1[0], c1[1], c1[2], B);
    order(e0, e1);
    idx = project_indices<false>(px, e0, e1, B, qf);

    out[0] = ((uint32_t)e1.code << 16) + (uint32_t)e0.code;
    // linear order {0,1,2,3} -> BC1 order {0,2,3,1}              [kernel.ispc:482-492]
    const uint32_t lo_bits = idx & 0x55555555u, hi_bits = idx & 0xAAAAAAAAu;
    out[1] = (hi_bits >> 1) + (hi_bits ^ (lo_bits << 1));
}

// Alpha part of BC3: min/max endpoints, 8-level ramp.            [kernel.ispc:535-571]
// Per texel the reference computes u = clamp((int)((a - lo) * scale + 0.5), 0, 7), then q = 7 - u reordered for DXT5
// (0 = alpha0 = max, 1 = alpha1 = min, 2.. the ramp): q' = q + 1 for q in 1..6, 0 for q = 0, 1 for q = 7.  Here u is taken in
// the float domain (the argument lies in [0.5, 7.6]: min with 7, floor), packed by Horner steps acc*8 + u on two 8-texel
// halves (exact: below 2^24) and the subtraction and the reordering run ONCE on the packed 3-bit fields.
__device__ __forceinline__ uint32_t dxt5_order_8x3(uint32_t u)     // eight 3-bit fields u -> q'
{
    const uint32_t ones = 011111111u;                              // octal: bit 0 of every field
    const uint32_t q = 077777777u - u;                             // 7 - u per field, no borrows
    const uint32_t any = (q | (q >> 1) | (q >> 2)) & ones;         // q != 0
    const uint32_t all = (q & (q >> 1) & (q >> 2)) & ones;         // q == 7
    return ((q & ~(all * 7u)) + (any & ~all)) | all;               // 7 -> 1, 0 -> 0, else + 1 (no carries: at most 6 + 1)
}
__device__ __forceinline__ void encode_alpha(const f2 (&a)[8], uint32_t out[2], const Bc1Tables& B)
{
    float lo = 255.f, hi = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j++) {                                   // minps / maxps of ordinary numbers (bytes)
        lo = __builtin_fminf(lo, a[j].x); hi = __builtin_fmaxf(hi, a[j].x);
        lo = __builtin_fminf(lo, a[j].y); hi = __builtin_fmaxf(hi, a[j].y);
    }
    if (lo == hi) hi = lo + 0.1f;
    const float scale = 7.0f * rcp_nr_pos(hi - lo, B);   // hi - lo in [0.1, 255]

    const f2 lo2 = {lo, lo}, sc2 = {scale, scale}, h2 = {0.5f, 0.5f};
    float half[2] = {0.f, 0.f};
#pragma unroll
    for (int j = 7; j >= 0; j--) {
        const f2 x = (a[j] - lo2) * sc2 + h2;                       // hi - lo >= 0.1: finite, in [0.5, 7.6]; two texels per instruction
#pragma unroll
        for (int t = 1; t >= 0; t--) {
            const float u = __builtin_floorf(vmin_raw(t ? x.y : x.x, 7.0f));
            float& h = half[j >> 2];
            h = h * 8.0f;
            h = h + u;
        }
    }
    const uint32_t q0 = dxt5_order_8x3((uint32_t)half[0]), q1 = dxt5_order_8x3((uint32_t)half[1]);   // 8 x 3 bits each
    out[0] = (uint32_t)(iclamp(cvt_i32_sat(lo), 0, 255) * 256 + iclamp(cvt_i32_sat(hi), 0, 255)) | (q0 << 16);
    out[1] = (q0 >> 16) | (q1 << 8);
}

// The sixteen texel words of block `cur` (4 x dwordx4 when VEC16)
template <bool VEC16>
__device__ __forceinline__ void load_words(uint32_t (&w)[16], const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t cur)
{
    const int32_t yy = cur / blocks_x, xx = cur - yy * blocks_x;
    const uint8_t* p = src + (int64_t)yy * 4 * stride + (int64_t)xx * 16;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        if (VEC16) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + y * stride);
            w[4 * y] = v.x; w[4 * y + 1] = v.y; w[4 * y + 2] = v.z; w[4 * y + 3] = v.w;
        } else {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p + y * stride);
            w[4 * y] = q[0]; w[4 * y + 1] = q[1]; w[4 * y + 2] = q[2]; w[4 * y + 3] = q[3];
        }
    }
}

// the same from a 32-bit byte offset (SMALL surfaces: every texel below 2 GiB from `src`, stride > 0): the uniform base stays in
// SGPRs, a lane carries one offset
template <bool VEC16>
__device__ __forceinline__ void load_words_at(uint32_t (&w)[16], const uint8_t* __restrict__ src, uint32_t off, uint32_t stride32)
{
#pragma unroll
    for (int y = 0; y < 4; y++) {
        const uint8_t* p = src + (off + (uint32_t)y * stride32);
        if (VEC16) {
            const uint4 v = *reinterpret_cast<const uint4*>(p);
            w[4 * y] = v.x; w[4 * y + 1] = v.y; w[4 * y + 2] = v.z; w[4 * y + 3] = v.w;
        } else {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
            w[4 * y] = q[0]; w[4 * y + 1] = q[1]; w[4 * y + 2] = q[2]; w[4 * y + 3] = q[3];
        }
    }
}

// SMALL (round 4; every surface below 2 GiB with a positive stride): a lane keeps its block's column and 32-bit byte offset and
// advances them by workgroup-uniform steps from chunk to chunk -- one division per workgroup instead of one per chunk, no 64-bit row
// pointers (v_mul_hi / v_mad_u64 / v_lshl_add_u64: 120 of the 2 820 issue cycles of a BC1 block-wave).
template <bool BC3, bool VEC16, bool SMALL>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BC13_WAVES, BC13_WAVES)))
bc13_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, uint8_t* __restrict__ dst)
{
    __shared__ __attribute__((aligned(16))) unsigned char s_tables[BC1_LDS_BYTES];
    // The first chunk's texels are requested BEFORE the tables are staged: at launch every wave of the chip stands in this
    // prologue at once, and the HBM round trip of the texels then runs under the 9 KiB table copy instead of after it.
    int32_t base = blockIdx.x * 256;
    uint32_t w[16];
    const uint32_t stride32 = (uint32_t)stride;
    const int32_t step_blocks = (int32_t)gridDim.x * 256;
    const int32_t step_y = step_blocks / blocks_x, step_x = step_blocks - step_y * blocks_x;       // scalar, once
    const uint32_t off_step = (uint32_t)step_y * 4u * stride32 + (uint32_t)step_x * 16u;
    const uint32_t off_wrap = 4u * stride32 - (uint32_t)blocks_x * 16u;                            // one block row down, blocks_x blocks back
    int32_t xx = 0;
    uint32_t off = 0;
    {
        const int32_t cur = base + threadIdx.x;
        const int32_t c0 = cur < nblocks ? cur : nblocks - 1;
        if (SMALL) {
            const int32_t yy = c0 / blocks_x;
            xx = c0 - yy * blocks_x;
            off = (uint32_t)yy * 4u * stride32 + (uint32_t)xx * 16u;
            load_words_at<VEC16>(w, src, off, stride32);
        } else {
            load_words<VEC16>(w, src, stride, blocks_x, c0);
        }
    }
    const Bc1Tables B = stage_bc1_tables(s_tables, threadIdx.x, 256);
    __syncthreads();
    for (;;) {
        const int32_t cur = base + threadIdx.x;
        // (requesting the NEXT chunk's texels before encoding this one was measured: the 16 extra live registers cost more
        // than the hidden latency gains -- four waves per SIMD already overlap each other's loads)
        Texels px;
        f2 al[8];
        // one v_cvt_f32_ubyteN per channel, as instructions: from C++ the compiler recognises sums of converted bytes as integer
        // sums and rebuilds them from v_bfe / v_add3 / v_cvt_f32_u32 (4-cycle forms) instead of 2-cycle float adds
#pragma unroll
        for (int k = 0; k < 16; k++) {
            px.rg[k].x = ubyte_f32<0>(w[k]);
            px.rg[k].y = ubyte_f32<1>(w[k]);
            if (k & 1) px.b2[k >> 1].y = ubyte_f32<2>(w[k]); else px.b2[k >> 1].x = ubyte_f32<2>(w[k]);
            if (BC3) { if (k & 1) al[k >> 1].y = ubyte_f32<3>(w[k]); else al[k >> 1].x = ubyte_f32<3>(w[k]); }
        }

        if (BC3) {
            uint32_t o[4];
            encode_alpha(al, &o[0], B);
            encode_color(px, &o[2], B);
            if (cur < nblocks) {
                uint32_t* d = reinterpret_cast<uint32_t*>(SMALL ? dst + (uint32_t)cur * 16u : dst + (int64_t)cur * 16);
                if (VEC16) *reinterpret_cast<uint4*>(d) = make_uint4(o[0], o[1], o[2], o[3]);
                else { d[0] = o[0]; d[1] = o[1]; d[2] = o[2]; d[3] = o[3]; }
            }
        } else {
            uint32_t o[2];
            encode_color(px, o, B);
            if (cur < nblocks) {
                uint32_t* d = reinterpret_cast<uint32_t*>(SMALL ? dst + (uint32_t)cur * 8u : dst + (int64_t)cur * 8);
                if (VEC16) *reinterpret_cast<uint2*>(d) = make_uint2(o[0], o[1]);
                else { d[0] = o[0]; d[1] = o[1]; }
            }
        }
        base += gridDim.x * 256;
        if (base >= nblocks) break;                                    // wave-uniform
        const int32_t nxt = base + threadIdx.x;
        if (SMALL) {
            xx += step_x; off += off_step;
            if (xx >= blocks_x) { xx -= blocks_x; off += off_wrap; }
            // lanes past the end (last chunk only) re-read the last block, like the clamped block index of the general path
            const int32_t ly = (nblocks - 1) / blocks_x;
            const uint32_t last = (uint32_t)ly * 4u * stride32 + (uint32_t)(nblocks - 1 - ly * blocks_x) * 16u;
            load_words_at<VEC16>(w, src, nxt < nblocks ? off : last, stride32);
        } else {
            load_words<VEC16>(w, src, stride, blocks_x, nxt < nblocks ? nxt : nblocks - 1);
        }
    }
}

// Persistent workgroups, each walking chunks of 256 blocks.  How many: 1 024 are resident at once (4 waves per SIMD); fewer, longer-lived
// workgroups stage the 9 KiB table image less often, more of them balance the tail better.  Round 6 sweep (tools/bc13_timing.py over
// tools/build_variant.sh builds, profiles/r06_bc13_grid_sweep.txt; us per launch BC1 / BC3):
//     workgroups     1024          2048 (round 3-5)   3072          6144          8192          one per chunk
//     4096^2         29.3 / 33.0   28.1-28.6 / 31.4-31.5   27.8 / 30.4   27.7 / 30.7   28.5 / 30.8   28.2-28.4 / 30.8
//     16384^2        393 / 461     378-380 / 439-441       375 / 437     371 / 428     370 / 432     387 / 442
// -> 3 072 up to 8 192 chunks (a 4096^2 surface has 4 096), 6 144 above: 0.340 / 0.345 of the HBM peak at 4096^2, 0.407 / 0.392 at 16384^2.
#ifndef BC13_GRID_WGS
#define BC13_GRID_WGS 0                       // 0 = the rule above; a number fixes the bound (tools/build_variant.sh sweeps)
#endif
static unsigned bc13_grid(int64_t n)
{
    const int64_t chunks = (n + 255) / 256;
    const int64_t bound = BC13_GRID_WGS > 0 ? (int64_t)BC13_GRID_WGS : (chunks <= 8192 ? 3072 : 6144);
    return (unsigned)(chunks < bound ? chunks : bound);
}

// SMALL: positive stride and every texel row below 2 GiB from the base (16384^2 RGBA8 is 1 GiB): 32-bit offsets.  The output offset is
// 32-bit on that path too: with rows that overlap (a positive stride below 4 * width, which the ABI tolerates) the block stream can
// outgrow the texels, so it is bounded separately.
static bool bc13_small(int64_t stride, int height, int64_t nblocks) { return stride > 0 && (int64_t)height * stride < ((int64_t)1 << 31) && nblocks * 16 < ((int64_t)1 << 32); }

// VEC16 requires: src base and stride multiples of 16, dst multiple of 16 (BC3) / 8 (BC1).
void launch_bc1(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st)
{
    const int bx = width / 4, by = height / 4;
    const int64_t n = (int64_t)bx * by;
    if (n <= 0) return;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0;
    const dim3 grid(bc13_grid(n)), blk(256);
    if (bc13_small(stride, height, n)) {
        if (vec) hipLaunchKernelGGL((bc13_kernel<false, true, true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
        else     hipLaunchKernelGGL((bc13_kernel<false, false, true>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
    } else {
        if (vec) hipLaunchKernelGGL((bc13_kernel<false, true, false>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
        else     hipLaunchKernelGGL((bc13_kernel<false, false, false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
    }
}

void launch_bc3(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st)
{
    const int bx = width / 4, by = height / 4;
    const int64_t n = (int64_t)bx * by;
    if (n <= 0) return;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const dim3 grid(bc13_grid(n)), blk(256);
    if (bc13_small(stride, height, n)) {
        if (vec) hipLaunchKernelGGL((bc13_kernel<true, true, true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
        else     hipLaunchKernelGGL((bc13_kernel<true, false, true>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
    } else {
        if (vec) hipLaunchKernelGGL((bc13_kernel<true, true, false>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
        else     hipLaunchKernelGGL((bc13_kernel<true, false, false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
    }
}

} // namespace itw
