// bc6h.hip -- BC6H (unsigned half) encoder kernel for gfx950 (MI355X).
//
// Replaces kernel.ispc:2039-3139 (CompressBlocksBC6H_ispc) behind CompressBlocksBC6H
// (ispc_texcomp.cpp:432-435).  The reference has no signed encoder (BC6H_SF16 routes
// to the same code, IntelPlugin.cpp:841); half bit patterns are consumed as integers.
//
// Mapping: one 4x4 block per lane; 128 B in (16 x RGBA16F, two dwordx4 loads per texel
// row), 16 B out.  Search structure follows the reference -- span gated mode
// selection, PCA-ranked two-region shapes, unclamped PCA fits, LS refinement -- built
// from the shared device routines of bcn_core.hpp.  Differences in organisation:
//   * the PCA ranking of the 32 shapes depends only on the block, not on the mode
//     being tried, so it is computed once per block (the reference recomputes it for
//     each of the up to seven two-region modes) and kept in LDS;
//   * neither do the line fits of a shape: the scan visits the ranked shapes once and
//     tries every two-region mode of the profile on each fit (slow profiles: 6 modes),
//     keeping one winner per mode in LDS; the winners are then refined and committed in
//     the reference's mode order.  Likewise the one-region fit serves modes 10..13;
//   * index selection: decoded palette values are integers <= 65535, so
//     (int)(((64-w)*a + w*b + 32)/64) = a + floor((w*(b-a) + 32)/64) exactly in fp32; the
//     true divide of the projection is formed from one reciprocal per region and two
//     FMAs (exactly rounded, see select_hdr);
//   * headers are packed from the format's bit-layout table (bc6h_layout.hpp, derived
//     from the decoder's mode descriptors), one compile-time specialisation per mode,
//     instead of the reference's per-mode arithmetic scatter (kernel.ispc:2392-2980).
// fp32 VALU bound; no MFMA-shaped work.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include "bcn_core.hpp"
#include "bc6h_layout.hpp"
#include "kernels.hpp"

namespace itw {

constexpr int TPB6 = 256;
constexpr float INV65535 = 1.0f / 65535.0f;     // ep/(256*256f-1) under fast-math
constexpr float INV31 = 1.0f / 31.0f;

struct HLane {
    TexF tex;                 // uf16-domain texels (plane 3 unused)
    float best_err;
    // best encoding so far, turned into bits once at the end of the kernel
    int32_t best_q[2][2][4];  // endpoint codes [region][endpoint][channel]
    uint32_t best_qb[2];
    int32_t best_shape, best_mode;   // best_shape < 0: one-region block
    float lo[3], hi[3];       // per-channel bounds
    float max_span;
    int32_t max_span_idx;
    int32_t mode, epb;
    int32_t qlo[3], qhi[3];   // endpoint clamp window in code space
    SeedTables T;
    int32_t* keys;            // LDS column, 32 entries: keys[i * TPB6]
    uint32_t* wins;           // LDS column, 6 winners x {err bits, shape, -, -}: wins[(4 * m + f) * TPB6]
    uint8_t* order;           // LDS column (slow profiles of the one-kernel path): the ranked list written out, order[c * TPB6] = shape of entry c
};

// ---- format data (kernel.ispc:2080-2125) -----------------------------------
__device__ __forceinline__ float span_of(int mode)
{
    // float expressions truncated to int, as the reference's `uniform int span = span_table[mode]` does
    switch (mode) {
    case 0:  return (float)(int)(0.9f * 65535.f / 64);
    case 1:  return (float)(int)(0.9f * 65535.f / 4);
    case 2:  return (float)(int)(0.8f * 65535.f / 256);
    case 5:  return (float)(int)(0.9f * 65535.f / 32);
    case 6:  return (float)(int)(0.9f * 65535.f / 16);
    case 9:  return 65535.f;
    case 10: return 65535.f;
    case 11: return (float)(int)(0.95f * 65535.f / 8);
    case 12: return (float)(int)(0.95f * 65535.f / 32);
    default: return 6.f;   // 13
    }
}

__device__ __forceinline__ int bits_of(int mode)
{
    switch (mode) {
    case 0: return 10; case 1: return 7; case 2: return 11; case 5: return 9; case 6: return 8; case 9: return 6;
    case 10: return 10; case 11: return 11; case 12: return 12; default: return 16;
    }
}

__device__ __forceinline__ int32_t code_to_uf16(uint32_t v, int32_t bits)             // kernel.ispc:2130-2137
{
    if (bits >= 15) return (int32_t)v;
    if (v == 0u) return 0;
    if (v == (1u << bits) - 1u) return 0xFFFF;
    return (int32_t)((v * 2u + 1u) << (15 - bits));
}

// quantise one endpoint pair to the lane's precision, clamp into the delta window, reconstruct. [2139-2169]
__device__ __forceinline__ void quant_pair(int32_t (&q)[2][4], float (&e)[2][4], const HLane& ln)
{
    const int32_t levels = 1 << ln.epb;
    for (int i = 0; i < 2; i++)
        for (int p = 0; p < 3; p++) {
            int32_t v = f2i_x86(e[i][p] * INV65535 * (float)(levels - 1) + 0.5f);
            v = iclamp(v, 0, levels - 1);
            v = iclamp(v, ln.qlo[p], ln.qhi[p]);
            q[i][p] = v;
            e[i][p] = (float)code_to_uf16((uint32_t)v, ln.epb);
        }
    q[0][3] = q[1][3] = 0;
}

// clamp window centred on the block's mid-range                                          [2302-2330]
__device__ __forceinline__ void set_window(HLane& ln, float span, int wide_channel)
{
    const int32_t levels = 1 << ln.epb;
    for (int p = 0; p < 3; p++) {
        float s = span;
        if (wide_channel >= 0) s *= (p == wide_channel) ? 2.0f : 1.0f;
        const float middle = (ln.lo[p] + ln.hi[p]) * 0.5f;
        const float b0 = middle - s * 0.5f, b1 = middle + s * 0.5f;
        ln.qlo[p] = iclamp(f2i_x86(b0 * INV65535 * (float)(levels - 1) + 0.5f), 0, levels - 1);
        ln.qhi[p] = iclamp(f2i_x86(b1 * INV65535 * (float)(levels - 1) + 0.5f), 0, levels - 1);
    }
}

// ---- header packing from the layout table -------------------------------------------------
template <int MODE>
__device__ __forceinline__ void pack_header(BlockBits& bb, const int32_t (&q)[2][2][4], uint32_t shape)
{
    constexpr Bc6hLayout L = BC6H_LAYOUT[MODE];
    constexpr int HEADER = L.two_regions ? 82 : 65;
    // field values: W = region 0 low endpoint (absolute); X, Y, Z absolute or delta against W
    uint32_t f[4][3];
    for (int ch = 0; ch < 3; ch++) {
        const uint32_t w = (uint32_t)q[0][0][ch];
        f[0][ch] = w;
        f[1][ch] = L.transformed ? (uint32_t)q[0][1][ch] - w : (uint32_t)q[0][1][ch];
        f[2][ch] = L.transformed ? (uint32_t)q[1][0][ch] - w : (uint32_t)q[1][0][ch];
        f[3][ch] = L.transformed ? (uint32_t)q[1][1][ch] - w : (uint32_t)q[1][1][ch];
    }
#pragma unroll
    for (int i = 0; i < HEADER; i++) {
        const int field = L.slot[i] >> 4, bit = L.slot[i] & 15;
        uint32_t src;
        if (field == 1)      src = (uint32_t)L.prefix;
        else if (field == 2) src = shape;
        else                 src = f[(field - 3) & 3][(field - 3) >> 2];
        bb.put(i, 1, (src >> bit) & 1u);
    }
}

__device__ __forceinline__ void pack_header_dyn(BlockBits& bb, int mode, const int32_t (&q)[2][2][4], uint32_t shape)
{
    switch (mode) {
    case 0: pack_header<0>(bb, q, shape); break;   case 1: pack_header<1>(bb, q, shape); break;
    case 2: pack_header<2>(bb, q, shape); break;   case 3: pack_header<3>(bb, q, shape); break;
    case 4: pack_header<4>(bb, q, shape); break;   case 5: pack_header<5>(bb, q, shape); break;
    case 6: pack_header<6>(bb, q, shape); break;   case 7: pack_header<7>(bb, q, shape); break;
    case 8: pack_header<8>(bb, q, shape); break;   case 9: pack_header<9>(bb, q, shape); break;
    case 10: pack_header<10>(bb, q, shape); break; case 11: pack_header<11>(bb, q, shape); break;
    case 12: pack_header<12>(bb, q, shape); break; default: pack_header<13>(bb, q, shape); break;
    }
}

__device__ __forceinline__ void store_bits(uint32_t (&out)[4], const BlockBits& bb)
{
    out[0] = (uint32_t)bb.lo; out[1] = (uint32_t)(bb.lo >> 32); out[2] = (uint32_t)bb.hi; out[3] = (uint32_t)(bb.hi >> 32);
}

// two-region block: anchor rule as BC7 mode 1, 3-bit indices                             [2982-3010]
__device__ __forceinline__ void emit_two_region(uint32_t (&out)[4], int32_t (&q)[2][2][4], const uint32_t (&qb)[2], int shape, int mode)
{
    const Shape sh = load_shape(shape);
    const int a1 = (int)(sh.anchors >> 4);
    uint32_t flips = 0;
    for (int j = 0; j < 2; j++) {
        const int k0 = (j == 0) ? 0 : a1;
        const uint32_t word = (k0 < 8) ? qb[0] : qb[1];
        if (((word >> (4 * (k0 & 7))) & 15u) >= 4u) {
            for (int p = 0; p < 4; p++) { const int32_t t = q[j][0][p]; q[j][0][p] = q[j][1][p]; q[j][1][p] = t; }
            flips |= subset_mask(sh, j);
        }
    }
    BlockBits bb;
    pack_header_dyn(bb, mode, q, (uint32_t)shape);
    int pos = 82;
    for (int k = 0; k < 16; k++) {
        uint32_t v = ((k < 8 ? qb[0] >> (4 * k) : qb[1] >> (4 * (k - 8))) & 15u);
        if ((flips >> k) & 1u) v = 7u - v;
        const int n = (k == 0) ? 2 : 3;
        bb.put(pos, n, v); pos += n;
    }
    bb.drop_bit(82 + 3 * a1 + 1);
    store_bits(out, bb);
}

// one-region block: 4-bit indices                                                         [3012-3031]
__device__ __forceinline__ void emit_one_region(uint32_t (&out)[4], int32_t (&q)[2][4], uint32_t (&qb)[2], int mode)
{
    if ((qb[0] & 15u) >= 8u) {
        for (int p = 0; p < 4; p++) { const int32_t t = q[0][p]; q[0][p] = q[1][p]; q[1][p] = t; }
        qb[0] = 0xFFFFFFFFu - qb[0];
        qb[1] = 0xFFFFFFFFu - qb[1];
    }
    int32_t q2[2][2][4];
    for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) { q2[0][i][p] = q[i][p]; q2[1][i][p] = 0; }
    BlockBits bb;
    pack_header_dyn(bb, mode, q2, 0u);
    int pos = 65;
    for (int k = 0; k < 16; k++) {
        const uint32_t v = ((k < 8 ? qb[0] >> (4 * k) : qb[1] >> (4 * (k - 8))) & 15u);
        const int n = (k == 0) ? 3 : 4;
        bb.put(pos, n, v); pos += n;
    }
    store_bits(out, bb);
}

// ---- index selection for uf16-domain texels (kernel.ispc:1133-1193 with 16-bit endpoints) ---------
// One region's endpoints as the decoder reconstructs them (integers in [0,65535], held as floats).
struct HSeg {
    float a[3], ba[3];        // endpoint 0; endpoint 1 - endpoint 0 (exact)
    float div, rdiv;          // sum (b-a)^2 accumulated in the reference's order; its IEEE reciprocal
};

__device__ __forceinline__ HSeg make_hseg(const float (&e)[2][4])
{
    HSeg s;
#pragma unroll
    for (int p = 0; p < 3; p++) { s.a[p] = e[0][p]; s.ba[p] = e[1][p] - e[0][p]; }
    s.div = sq(s.ba[0]);                           // the reference's 0 + b0*b0: a square is never -0
    s.div += sq(s.ba[1]);
    s.div += sq(s.ba[2]);
    s.rdiv = 1.0f / s.div;
    return s;
}

// Returns the block error (sum of per-texel errors, each truncated with cvttps2dq semantics); indices in qb.
//  * `proj /= div` of the reference is a true divide.  Here q0 = RN(proj * r), r = RN(1/div); rem = proj - q0*div
//    (exact, one FMA); q = RN(q0 + rem*r) = RN(proj/div) (Markstein's correction step with a correctly rounded
//    reciprocal; additionally checked against the divide instruction for all 2^23 divisor significands x 400
//    numerators, 3.4e9 pairs, 0 mismatches).  div == 0: 0*inf = NaN on both routes -> index 1.
//  * decode: (64-w)*a, w*b <= 64*65535 and their sum + 32 are exact in fp32, /64 is exact, truncation of a
//    non-negative value is floor: d = a + floor((w*(b-a) + 32) / 64), every step exact (|w*(b-a)| < 2^23).
//  * float -> int of the index: cvttps2dq gives INT_MIN (-> clamp 1) for NaN and for x >= 2^31, where
//    v_cvt_i32_f32 would saturate upwards; x < -2^31 ends at 1 on both.
//  * WANT_IDX = false (the shape scans): only the error is produced.  Which of the two levels won and the packed indices
//    matter for a mode's winner alone, and finish_two_region recomputes them from the winner's endpoints.  Errors are
//    finite (texels and decoded endpoints are finite integers-as-floats), so the reference's `err0 < err1 ? err0 : err1`
//    is the IEEE minimum.
template <int BITS, int PAIRS, bool WANT_IDX = true>
__device__ __forceinline__ float select_hdr(uint32_t (&qb)[2], const TexF& px, const HSeg (&sg)[2], uint32_t pattern)
{
    constexpr int LEVELS = 1 << BITS;
    float total = 0.f;
    qb[0] = qb[1] = 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        HSeg s = sg[0];
        if (PAIRS == 2) {
            const bool one = ((pattern >> (2 * k)) & 3u) == 1u;
#pragma unroll
            for (int p = 0; p < 3; p++) { s.a[p] = one ? sg[1].a[p] : s.a[p]; s.ba[p] = one ? sg[1].ba[p] : s.ba[p]; }
            s.div = one ? sg[1].div : s.div; s.rdiv = one ? sg[1].rdiv : s.rdiv;
        }
        float t[3];
#pragma unroll
        for (int p = 0; p < 3; p++) t[p] = px.get(p, k);
        // sums start from their first term: 0 + x differs from x only for x = -0, and a zero of either sign in `proj` gives
        // x = 0.5 (or the same NaN when rdiv is inf) below; the error sums start from squares, which are never -0
        float proj = (t[0] - s.a[0]) * s.ba[0];
#pragma unroll
        for (int p = 1; p < 3; p++) proj += (t[p] - s.a[p]) * s.ba[p];
        float q = proj * s.rdiv;
        const float rem = __builtin_fmaf(-q, s.div, proj);
        q = __builtin_fmaf(rem, s.rdiv, q);
        const float x = q * (float)LEVELS + 0.5f;
        int32_t q1 = iclamp((int32_t)x, 1, LEVELS - 1);
        q1 = (x < 2147483648.0f) ? q1 : 1;
        // the format's weights (q*128 + D) / (2*D), D = LEVELS - 1, = floor(q*64/D + 1/2): the fractional parts stay at least
        // 0.03 away from an integer for every q of both index widths (checked in tests/test_exact_forms.py), so the float
        // form is exact and costs a convert, two FMAs and two floors instead of two integer divisions and two converts
        const float fq = (float)q1;
        constexpr float STEP = 64.0f / (float)(LEVELS - 1);
        const float w1 = __builtin_floorf(__builtin_fmaf(fq, STEP, 0.5f));
        const float w0 = __builtin_floorf(__builtin_fmaf(fq, STEP, 0.5f - STEP));
        float err0, err1;
#pragma unroll
        for (int p = 0; p < 3; p++) {
            const float d0 = s.a[p] + __builtin_floorf((w0 * s.ba[p] + 32.0f) * 0.015625f);
            const float d1 = s.a[p] + __builtin_floorf((w1 * s.ba[p] + 32.0f) * 0.015625f);
            if (p == 0) { err0 = sq(d0 - t[p]); err1 = sq(d1 - t[p]); }
            else        { err0 += sq(d0 - t[p]); err1 += sq(d1 - t[p]); }
        }
        float e;
        if (WANT_IDX) {
            const bool first = err0 < err1;
            e = first ? err0 : err1;
            const uint32_t qi = (uint32_t)(first ? q1 - 1 : q1);
            if (k < 8) qb[0] += qi << (4 * k); else qb[1] += qi << (4 * (k - 8));
        } else {
            e = __builtin_fminf(err0, err1);
        }
        // (float)cvttps2dq(e) for a finite e >= 0: floor(e) below 2^31, INT_MIN above
        total += (e < 2147483648.0f) ? __builtin_floorf(e) : -2147483648.0f;
    }
    return total;
}

// ---- mode descriptors ------------------------------------------------------------------------------
// gate a mode on the block's widest channel span and make it the lane's current mode          [2332-2365]
__device__ __forceinline__ bool enter_mode(HLane& ln, int mode, float margin)
{
    const float span = span_of(mode);
    if (ln.max_span * margin > span) return false;
    ln.epb = bits_of(mode);
    if (mode >= 10 || mode <= 1 || mode == 5 || mode == 9) {
        ln.mode = mode;
        set_window(ln, span, -1);
    } else {
        ln.mode = mode + ln.max_span_idx;           // 2 -> 2/3/4, 6 -> 6/7/8 by the widest channel
        set_window(ln, span, ln.max_span_idx);
    }
    return true;
}

// ---- searches ----------------------------------------------------------------------------------
__device__ __forceinline__ void rank_shapes32(HLane& ln)                                  // [2259-2269]
{
    Stats<3> full;
    stats_of<3>(full, ln.tex, 0xffffu);
    for (int part = 0; part < 32; part++) {
        ln.tex.fence();                             // texel products are loop invariant: do not hoist 100+ values
        const uint32_t m0 = BCN_SUBSET_MASKS[part] & 0xffffu;
        const int32_t bound = split_bound<3, true>(ln.tex, m0, full, ln.T);
        ln.keys[part * TPB6] = (int32_t)((uint32_t)part + (uint32_t)bound * 64u);
    }
}

__device__ __forceinline__ int32_t next_key32(const HLane& ln, int32_t prev, bool first)
{
    int32_t cur = 0x7fffffff;
    for (int i = 0; i < 32; i++) {
        const int32_t k = ln.keys[i * TPB6];
        if ((first || k > prev) && k <= cur) cur = k;
    }
    return cur;
}

// The first `count` entries of the ranked list, written out: after this the key column is dead (bc6h_kernel<true> reuses it for the winners).
__device__ __forceinline__ void order_shapes32(HLane& ln, int count)
{
    int32_t prev = 0;
    for (int c = 0; c < count; c++) {
        prev = next_key32(ln, prev, c == 0);
        ln.order[c * TPB6] = (uint8_t)(prev & 31);
    }
}

// The two-region modes a profile encodes, in the reference's order.  Slow profiles: 0,1,2,5,6,9 (never gated out:
// margin 0).  Other profiles: the mode the gate sequence left in the lane (`gated`), then mode 1 unless fast_mode.
__device__ __forceinline__ int two_region_mode(bool slow, int m, int gated)
{
    if (!slow) return (m == 0) ? gated : 1;
    return (m == 0) ? 0 : (m == 1) ? 1 : (m == 2) ? 2 : (m == 3) ? 5 : (m == 4) ? 6 : 9;
}

// Scan of the `count` best ranked shapes: one pair of line fits per shape serves all `nmodes` modes. [2174-2273]
// LEAN: the scan keeps errors only and the finish recomputes the winner's indices (pays when many candidates are scanned:
// the slow profiles); otherwise the scan stores the indices with the winner.
// `first`/`step`: the share of a split scan (wide path): list entries first, first + step, ... ; a winner also records
// its position in the ranked list (the reference's strict `<` keeps the earliest entry among equal errors).
template <bool LEAN, bool ORDERED = false>
__device__ __forceinline__ void scan_two_region(HLane& ln, bool slow, int nmodes, int gated, int count, int first = 0, int step = 1)
{
    for (int m = 0; m < nmodes; m++) {
        ln.wins[(4 * m + 0) * TPB6] = 0x7f800000u;      // +inf
        ln.wins[(4 * m + 1) * TPB6] = 0u;
        ln.wins[(4 * m + 2) * TPB6] = 0u;
        ln.wins[(4 * m + 3) * TPB6] = 0u;
    }
    int32_t prev = 0;
    for (int c = 0; c < count; c++) {
        prev = ORDERED ? (int32_t)ln.order[c * TPB6] : next_key32(ln, prev, c == 0);      // only the shape (low 5 bits) is used below
        if (step > 1 && (c % step) != first) continue;           // another wave's share of the list
        ln.tex.fence();
        const int shape = prev & 31;
        const Shape sh = load_shape(shape);
        float fit[2][2][4];
#pragma unroll
        for (int j = 0; j < 2; j++) fit_subset<3, false, true>(fit[j], ln.tex, subset_mask(sh, j), ln.T);
#pragma unroll 1
        for (int m = 0; m < nmodes; m++) {
            ln.tex.fence();
            enter_mode(ln, two_region_mode(slow, m, gated), 0.f);
            HSeg sg[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float ep[2][4];
                int32_t q[2][4];
#pragma unroll
                for (int i = 0; i < 2; i++) for (int p = 0; p < 3; p++) ep[i][p] = fit[j][i][p];
                quant_pair(q, ep, ln);
                sg[j] = make_hseg(ep);
            }
            uint32_t qb[2];
            const float err = select_hdr<3, 2, !LEAN>(qb, ln.tex, sg, sh.pattern);
            if (err < __uint_as_float(ln.wins[(4 * m + 0) * TPB6])) {
                ln.wins[(4 * m + 0) * TPB6] = __float_as_uint(err);
                ln.wins[(4 * m + 1) * TPB6] = (uint32_t)shape;
                if (!LEAN) { ln.wins[(4 * m + 2) * TPB6] = qb[0]; ln.wins[(4 * m + 3) * TPB6] = qb[1]; }
                else ln.wins[(4 * m + 2) * TPB6] = (uint32_t)c;   // list position (LEAN winners carry no indices)
            }
        }
    }
}

// Refinement of each mode's winner, then the mode competes for the block, in mode order.
template <bool LEAN>
__device__ __forceinline__ void finish_two_region(HLane& ln, bool slow, int nmodes, int gated, int refine, int m0 = 0)
{
#pragma unroll 1
    for (int m = m0; m < nmodes; m++) {
        ln.tex.fence();
        enter_mode(ln, two_region_mode(slow, m, gated), 0.f);
        float berr = __uint_as_float(ln.wins[(4 * m + 0) * TPB6]);
        const int bshape = (int)ln.wins[(4 * m + 1) * TPB6];
        uint32_t bqb[2] = {ln.wins[(4 * m + 2) * TPB6], ln.wins[(4 * m + 3) * TPB6]};
        const Shape sh = load_shape(bshape);
        // endpoint codes and indices of the scan's winner (the scan kept only its error): same fit, same quantiser,
        // same bits, one selection pass
        int32_t bq[2][2][4];
        {
            HSeg sg0[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float ep[2][4];
                fit_subset<3, false, true>(ep, ln.tex, subset_mask(sh, j), ln.T);
                quant_pair(bq[j], ep, ln);
                sg0[j] = make_hseg(ep);
            }
            if (LEAN) (void)select_hdr<3, 2>(bqb, ln.tex, sg0, sh.pattern);
        }
        if (berr == __builtin_inff()) {             // no candidate beat +inf (cannot happen: errors are finite); keep zeros
            for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) bq[j][i][p] = 0;
            bqb[0] = bqb[1] = 0u;
        }
        for (int it = 0; it < refine; it++) {
            ln.tex.fence();
            HSeg sg[2];
            int32_t q[2][2][4];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                float ep[2][4];
                refit_subset<3, 3>(ep, ln.tex, bqb, subset_mask(sh, j), ln.T);
                quant_pair(q[j], ep, ln);
                sg[j] = make_hseg(ep);
            }
            uint32_t qb[2];
            const float err = select_hdr<3, 2>(qb, ln.tex, sg, sh.pattern);
            if (err < berr) {
                for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) bq[j][i][p] = q[j][i][p];
                bqb[0] = qb[0]; bqb[1] = qb[1];
                berr = err;
            }
        }
        if (berr < ln.best_err) {
            ln.best_err = berr;
            for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) ln.best_q[j][i][p] = bq[j][i][p];
            ln.best_qb[0] = bqb[0]; ln.best_qb[1] = bqb[1];
            ln.best_shape = bshape; ln.best_mode = ln.mode;
        }
    }
}

// one-region modes: the fit is the block's, whatever the mode                                   [2275-2300]
// slow: modes 10..13 in turn; otherwise the one mode the gates left in the lane.
__device__ __forceinline__ void encode_one_region(HLane& ln, bool slow, int refine, int m0 = 0, int m1 = 4)
{
    float fit[2][4];
    fit_subset<3, false, true>(fit, ln.tex, 0xffffu, ln.T);
    const int n = slow ? m1 : 1;
#pragma unroll 1
    for (int m = slow ? m0 : 0; m < n; m++) {
        ln.tex.fence();
        if (slow) enter_mode(ln, 10 + m, 0.f);
        float ep[2][4];
        int32_t q[2][4];
        uint32_t qb[2];
        HSeg sg[2];
#pragma unroll
        for (int i = 0; i < 2; i++) for (int p = 0; p < 3; p++) ep[i][p] = fit[i][p];
        quant_pair(q, ep, ln);
        sg[0] = make_hseg(ep); sg[1] = sg[0];
        float err = select_hdr<4, 1>(qb, ln.tex, sg, 0u);
        for (int it = 0; it < refine; it++) {
            refit_subset<4, 3>(ep, ln.tex, qb, 0xffffu, ln.T);
            quant_pair(q, ep, ln);
            sg[0] = make_hseg(ep);
            err = select_hdr<4, 1>(qb, ln.tex, sg, 0u);
        }
        if (err < ln.best_err) {
            ln.best_err = err;
            for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) { ln.best_q[0][i][p] = q[i][p]; ln.best_q[1][i][p] = 0; }
            ln.best_qb[0] = qb[0]; ln.best_qb[1] = qb[1];
            ln.best_shape = -1; ln.best_mode = ln.mode;
        }
    }
}

// texels -> uf16 code space, channel bounds, widest channel, empty best                 [kernel.ispc:134-151, 3036-3067]
template <bool VEC16>
__device__ __forceinline__ void load_and_setup(HLane& ln, const uint8_t* __restrict__ src, int64_t stride, int32_t xx, int32_t yy)
{
    // load: 4 texels x 8 bytes per row; keep R,G,B half bit patterns as integers    [kernel.ispc:134-151]
    const uint8_t* p = src + (int64_t)yy * 4 * stride + (int64_t)xx * 32;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        uint32_t w[8];
        if (VEC16) {
            const uint4 v0 = *reinterpret_cast<const uint4*>(p + y * stride);
            const uint4 v1 = *reinterpret_cast<const uint4*>(p + y * stride + 16);
            w[0] = v0.x; w[1] = v0.y; w[2] = v0.z; w[3] = v0.w; w[4] = v1.x; w[5] = v1.y; w[6] = v1.z; w[7] = v1.w;
        } else {
            const uint16_t* q = reinterpret_cast<const uint16_t*>(p + y * stride);
            for (int i = 0; i < 8; i++) w[i] = (uint32_t)q[2 * i] | ((uint32_t)q[2 * i + 1] << 16);
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
            ln.tex.v[0][y * 4 + x] = (float)(w[2 * x] & 0xffffu);
            ln.tex.v[1][y * 4 + x] = (float)(w[2 * x] >> 16);
            ln.tex.v[2][y * 4 + x] = (float)(w[2 * x + 1] & 0xffffu);
            ln.tex.v[3][y * 4 + x] = 0.f;
        }
    }

    // half bits -> uf16 code space (x/31*64), channel bounds, widest channel          [kernel.ispc:3036-3067]
    for (int c = 0; c < 3; c++) { ln.lo[c] = 65535.f; ln.hi[c] = 0.f; }
    for (int c = 0; c < 3; c++)
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float v = (ln.tex.v[c][k] * INV31) * 64.0f;
            ln.tex.v[c][k] = v;
            ln.lo[c] = fmin_x86(ln.lo[c], v);
            ln.hi[c] = fmax_x86(ln.hi[c], v);
        }
    ln.max_span = 0.f;
    ln.max_span_idx = 0;
    for (int c = 0; c < 3; c++) {
        const float s = ln.hi[c] - ln.lo[c];
        if (s > ln.max_span) { ln.max_span_idx = c; ln.max_span = s; }
    }

    ln.best_err = __builtin_inff();
    for (int j = 0; j < 2; j++) for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) ln.best_q[j][i][p] = 0;
    ln.best_qb[0] = ln.best_qb[1] = 0u;
    ln.best_shape = -1; ln.best_mode = 10;
    ln.mode = 0; ln.epb = 0;
    for (int c = 0; c < 3; c++) { ln.qlo[c] = 0; ln.qhi[c] = 0; }
}

// Register / LDS budgets (per 256-lane workgroup; a CU has 160 KiB, so 53 KiB is what a third workgroup = a third wave per SIMD needs).
//   slow profiles     seed tables 12 KiB + 32 keys 32 KiB + the ranked list written out 8 KiB; the 6 winners (24 KiB) live in the key column, dead
//                     once the list is written: 52 KiB
//   other profiles    seed tables + 32 keys + at most 2 winners (8 KiB): 52 KiB
//   no two-region search (`veryfast`): bc6h_one_region_kernel, seed tables only
// BC6H_*_WAVES = the occupancy an instantiation is compiled for; measured on MI355X (profiles/r06_waves_sweep.txt): the kernels are bound by
// dependent issue and LDS latency, a third wave with 140 spilled registers beats two waves without by 9-15 %.
#ifndef BC6H_SLOW_WAVES
#define BC6H_SLOW_WAVES 3
#endif
#ifndef BC6H_WIDE_WAVES
#define BC6H_WIDE_WAVES 3
#endif
#ifndef BC6H_FAST_WAVES
#define BC6H_FAST_WAVES 3
#endif
#ifndef BC6H_ONE_WAVES
#define BC6H_ONE_WAVES 4
#endif
constexpr int bc6h_waves(bool slow) { return slow ? BC6H_SLOW_WAVES : BC6H_FAST_WAVES; }
template <bool SLOW, bool VEC16>
__global__ void __launch_bounds__(TPB6) __attribute__((amdgpu_waves_per_eu(bc6h_waves(SLOW), bc6h_waves(SLOW))))
bc6h_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks,
            uint8_t* __restrict__ dst, const bc6h_enc_settings S)
{
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    __shared__ int32_t s_keys[32 * TPB6];
    __shared__ uint32_t s_tail[8 * TPB6];            // slow: the ranked list (bytes, 32 rows); otherwise the winners' columns (2 winners x 4 words)
    HLane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB6);
    __syncthreads();
    const int32_t gid = blockIdx.x * TPB6 + threadIdx.x;
    const bool live = gid < nblocks;
    const int32_t b = live ? gid : nblocks - 1;    // idle lanes of the last workgroup redo its last block, store nothing
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    ln.keys = s_keys + threadIdx.x;
    ln.wins = SLOW ? reinterpret_cast<uint32_t*>(s_keys) + threadIdx.x : s_tail + threadIdx.x;
    ln.order = reinterpret_cast<uint8_t*>(s_tail) + threadIdx.x;

    load_and_setup<VEC16>(ln, src, stride, xx, yy);

    if (SLOW) {                                                                         // [kernel.ispc:3073-3085]
        // every mode is encoded (margin 0 never gates); the ranking runs even when no shape is tried
        rank_shapes32(ln);
        const int count = min(max(S.fastSkipTreshold, 0), 32);
        if (count > 0) {
            order_shapes32(ln, count);              // the keys are dead from here: the winners' columns take their place
            scan_two_region<true, true>(ln, true, 6, 0, count);
            finish_two_region<true>(ln, true, 6, 0, S.refineIterations_2p);
        }
        encode_one_region(ln, true, S.refineIterations_1p);
    } else {                                                                            // [kernel.ispc:3086-3106]
        const float inv1_2 = 1.0f / 1.2f;
        if (S.fastSkipTreshold > 0) {
            enter_mode(ln, 9, 0.f);
            if (S.fast_mode) enter_mode(ln, 1, 1.f);
            enter_mode(ln, 6, inv1_2);
            enter_mode(ln, 5, inv1_2);
            enter_mode(ln, 0, inv1_2);
            enter_mode(ln, 2, 1.f);
            // the gates leave one of 9,1,6,5,0,2 in the lane; recover the base mode (6/7/8 -> 6, 2/3/4 -> 2)
            const int gated = (ln.mode >= 6 && ln.mode <= 8) ? 6 : ((ln.mode >= 2 && ln.mode <= 4) ? 2 : ln.mode);
            rank_shapes32(ln);
            const int count = min(S.fastSkipTreshold, 32);
            const int nmodes = S.fast_mode ? 1 : 2;
            scan_two_region<false>(ln, false, nmodes, gated, count);
            finish_two_region<false>(ln, false, nmodes, gated, S.refineIterations_2p);
        }
        enter_mode(ln, 10, 0.f);
        enter_mode(ln, 11, 1.f);
        enter_mode(ln, 12, 1.f);
        enter_mode(ln, 13, 1.f);
        encode_one_region(ln, false, S.refineIterations_1p);
    }

    uint32_t out[4];
    if (ln.best_shape >= 0) emit_two_region(out, ln.best_q, ln.best_qb, ln.best_shape, ln.best_mode);
    else                    emit_one_region(out, ln.best_q[0], ln.best_qb, ln.best_mode);
    if (live) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (int64_t)b * 16);
        if (VEC16) *reinterpret_cast<uint4*>(d) = make_uint4(out[0], out[1], out[2], out[3]);
        else { d[0] = out[0]; d[1] = out[1]; d[2] = out[2]; d[3] = out[3]; }
    }
}

// Profiles without a two-region search (fastSkipTreshold <= 0 and not slow_mode: `veryfast`): gated modes 10..13 only.  No ranking, no winners:
// the seed tables are all the LDS it needs, and it keeps the register budget of two waves per SIMD (at three it spills and loses 24 %).
template <bool VEC16>
__global__ void __launch_bounds__(TPB6) __attribute__((amdgpu_waves_per_eu(BC6H_ONE_WAVES, BC6H_ONE_WAVES)))
bc6h_one_region_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks,
                       uint8_t* __restrict__ dst, const bc6h_enc_settings S)
{
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    HLane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB6);
    __syncthreads();
    const int32_t gid = blockIdx.x * TPB6 + threadIdx.x;
    const bool live = gid < nblocks;
    const int32_t b = live ? gid : nblocks - 1;
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    ln.keys = nullptr;
    ln.wins = nullptr;
    ln.order = nullptr;

    load_and_setup<VEC16>(ln, src, stride, xx, yy);
    enter_mode(ln, 10, 0.f);                                                            // [kernel.ispc:3100-3106]
    enter_mode(ln, 11, 1.f);
    enter_mode(ln, 12, 1.f);
    enter_mode(ln, 13, 1.f);
    encode_one_region(ln, false, S.refineIterations_1p);

    uint32_t out[4];
    emit_one_region(out, ln.best_q[0], ln.best_qb, ln.best_mode);
    if (live) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (int64_t)b * 16);
        if (VEC16) *reinterpret_cast<uint4*>(d) = make_uint4(out[0], out[1], out[2], out[3]);
        else { d[0] = out[0]; d[1] = out[1]; d[2] = out[2]; d[3] = out[3]; }
    }
}

// ---- WIDE path for the slow profiles: calls too small to fill the chip (see bc7.hip, same idea) ---------------------------
//   phase A  blockIdx.y = task: `parts` strided shares of the ranked two-region list (every share ranks the 32 shapes
//            itself: the ranking is a tenth of the scan), each leaving one winner {error, shape, list position} per mode;
//            next to them one task per one-region mode (10..13), which depends on nothing;
//   phase B  blockIdx.y = two-region mode: ordered argmin over the shares (lowest error, then earliest list position: the
//            reference's strict `<`, kernel.ispc:2215), refit + refine, emit the mode's candidate block;
//   commit   candidates compete in the reference's order 0,1,2,5,6,9,10,11,12,13 with strict `<` from +inf
//            (kernel.ispc:3073-3085).
// Same device functions, same arithmetic, same bytes as the one-kernel path; ITW_BC6H_PATH=deep|wide forces either.
constexpr int W6_MAX_PARTS = 8;
constexpr int W6_SLOTS = 10;              // 6 two-region modes + 4 one-region modes

template <bool VEC16>
__global__ void __launch_bounds__(TPB6) __attribute__((amdgpu_waves_per_eu(BC6H_WIDE_WAVES, BC6H_WIDE_WAVES)))
bc6h_wide_phaseA(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, uint4* __restrict__ wins,
                 float* __restrict__ cerr, uint4* __restrict__ cblk, const bc6h_enc_settings S, const int parts, const int count)
{
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    __shared__ int32_t s_keys[32 * TPB6];           // the winners' columns take the keys' place once the list is written out (bc6h_kernel<true>)
    __shared__ uint8_t s_order[32 * TPB6];
    HLane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB6);
    __syncthreads();
    const int32_t gid = blockIdx.x * TPB6 + threadIdx.x;
    const bool live = gid < nblocks;
    const int32_t b = live ? gid : nblocks - 1;
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    ln.keys = s_keys + threadIdx.x;
    ln.wins = reinterpret_cast<uint32_t*>(s_keys) + threadIdx.x;
    ln.order = s_order + threadIdx.x;
    load_and_setup<VEC16>(ln, src, stride, xx, yy);
    const int task = blockIdx.y;                                   // wave-uniform
    if (task < parts) {                                            // a share of the two-region scan
        rank_shapes32(ln);
        order_shapes32(ln, count);
        scan_two_region<true, true>(ln, true, 6, 0, count, task, parts);
        if (live)
            for (int m = 0; m < 6; m++)
                wins[((int64_t)m * parts + task) * nblocks + b] = make_uint4(ln.wins[(4 * m + 0) * TPB6], ln.wins[(4 * m + 1) * TPB6], ln.wins[(4 * m + 2) * TPB6], 0u);
    } else {                                                       // one one-region mode
        const int m = task - parts;
        encode_one_region(ln, true, S.refineIterations_1p, m, m + 1);
        uint32_t out[4] = {0u, 0u, 0u, 0u};
        if (ln.best_err < __builtin_inff()) emit_one_region(out, ln.best_q[0], ln.best_qb, ln.best_mode);
        if (live) { cerr[(int64_t)(6 + m) * nblocks + b] = ln.best_err; cblk[(int64_t)(6 + m) * nblocks + b] = make_uint4(out[0], out[1], out[2], out[3]); }
    }
}

template <bool VEC16>
__global__ void __launch_bounds__(TPB6) __attribute__((amdgpu_waves_per_eu(BC6H_WIDE_WAVES, BC6H_WIDE_WAVES)))
bc6h_wide_phaseB(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, const uint4* __restrict__ wins,
                 float* __restrict__ cerr, uint4* __restrict__ cblk, const bc6h_enc_settings S, const int parts)
{
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    __shared__ uint32_t s_wins[24 * TPB6];
    HLane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB6);
    __syncthreads();
    const int32_t gid = blockIdx.x * TPB6 + threadIdx.x;
    const bool live = gid < nblocks;
    const int32_t b = live ? gid : nblocks - 1;
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    ln.keys = nullptr;
    ln.wins = s_wins + threadIdx.x;
    ln.order = nullptr;
    load_and_setup<VEC16>(ln, src, stride, xx, yy);
    const int m = blockIdx.y;                                      // wave-uniform: index into 0,1,2,5,6,9
    // ordered argmin over the shares: lowest error, then earliest list position (errors compare as floats, like the scan)
    uint4 best = wins[((int64_t)m * parts) * nblocks + b];
    for (int p = 1; p < parts; p++) {
        const uint4 x = wins[((int64_t)m * parts + p) * nblocks + b];
        const float ex = __uint_as_float(x.x), eb = __uint_as_float(best.x);
        if (ex < eb || (ex == eb && ex < __builtin_inff() && x.z < best.z)) best = x;
    }
    ln.wins[(4 * m + 0) * TPB6] = best.x; ln.wins[(4 * m + 1) * TPB6] = best.y;
    ln.wins[(4 * m + 2) * TPB6] = 0u; ln.wins[(4 * m + 3) * TPB6] = 0u;
    finish_two_region<true>(ln, true, m + 1, 0, S.refineIterations_2p, m);
    uint32_t out[4] = {0u, 0u, 0u, 0u};
    if (ln.best_err < __builtin_inff()) emit_two_region(out, ln.best_q, ln.best_qb, ln.best_shape, ln.best_mode);
    if (live) { cerr[(int64_t)m * nblocks + b] = ln.best_err; cblk[(int64_t)m * nblocks + b] = make_uint4(out[0], out[1], out[2], out[3]); }
}

__global__ void __launch_bounds__(TPB6)
bc6h_wide_commit(const float* __restrict__ cerr, const uint4* __restrict__ cblk, int32_t nblocks, uint32_t active, uint4* __restrict__ dst, int vec16)
{
    const int32_t b = blockIdx.x * TPB6 + threadIdx.x;
    if (b >= nblocks) return;
    float best = __builtin_inff();
    uint4 blk = make_uint4(0u, 0u, 0u, 0u);
    bool any = false;
    for (int slot = 0; slot < W6_SLOTS; slot++) {
        if (!((active >> slot) & 1u)) continue;
        const float e = cerr[(int64_t)slot * nblocks + b];
        if (e < best) { best = e; blk = cblk[(int64_t)slot * nblocks + b]; any = true; }
    }
    if (!any) {
        // nothing beat +inf (only with NaN errors): the one-kernel path then emits its initial state, an all-zero mode 10 block
        int32_t q[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
        uint32_t qb[2] = {0u, 0u}, out[4];
        emit_one_region(out, q, qb, 10);
        blk = make_uint4(out[0], out[1], out[2], out[3]);
    }
    if (vec16) dst[b] = blk;
    else { uint32_t* d = reinterpret_cast<uint32_t*>(dst) + (int64_t)b * 4; d[0] = blk.x; d[1] = blk.y; d[2] = blk.z; d[3] = blk.w; }
}

#ifndef ITW_BC6H_WIDE_MAX_BLOCKS
#define ITW_BC6H_WIDE_MAX_BLOCKS 98304       // measured crossover on MI355X (tools/bc7_path_probe.py --bc6h)
#endif
static std::atomic<int> g_bc6h_path{-1};      // 0 by size, 1 deep, 2 wide
static int bc6h_path_override()
{
    int v = g_bc6h_path.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = std::getenv("ITW_BC6H_PATH");
        v = !e ? 0 : !std::strcmp(e, "deep") ? 1 : !std::strcmp(e, "wide") ? 2 : 0;
        g_bc6h_path.store(v, std::memory_order_relaxed);
    }
    return v;
}
void set_bc6h_path(int v) { g_bc6h_path.store(v == 1 || v == 2 ? v : 0, std::memory_order_relaxed); }

static bool bc6h_use_wide(int64_t n, const bc6h_enc_settings& s)
{
    if (!s.slow_mode || s.fastSkipTreshold < 2) return false;      // the fast profiles try two to four shapes: nothing to split
    const int o = bc6h_path_override();
    if (o == 1) return false;
    if (o == 2) return n <= ((int64_t)1 << 20);
    return n <= ITW_BC6H_WIDE_MAX_BLOCKS;
}
// wide: winners [6 modes][parts] x 16 B + candidates [10 slots] x (4 + 16) B per block; the one-kernel path needs nothing
size_t bc6h_workspace_bytes(int width, int height, const bc6h_enc_settings& s)
{
    const size_t n = (size_t)(width / 4) * (size_t)(height / 4);
    if (!bc6h_use_wide((int64_t)n, s)) return 0;
    return (size_t)6 * W6_MAX_PARTS * n * sizeof(uint4) + (((size_t)W6_SLOTS * n * sizeof(float) + 15) & ~(size_t)15) + (size_t)W6_SLOTS * n * sizeof(uint4);
}

static void launch_bc6h_wide(const uint8_t* src, int64_t stride, int bx, int64_t n, uint8_t* dst, const bc6h_enc_settings& s, void* workspace, hipStream_t st)
{
    uint4* wins = reinterpret_cast<uint4*>(workspace);
    float* cerr = reinterpret_cast<float*>(wins + (size_t)6 * W6_MAX_PARTS * n);
    uint4* cblk = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(cerr) + (((size_t)W6_SLOTS * n * sizeof(float) + 15) & ~(size_t)15));
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const int count = s.fastSkipTreshold > 32 ? 32 : s.fastSkipTreshold;
    // shares of the ranked list: while all waves (BC6H_WIDE_WAVES per SIMD) stay resident at once; every share repeats the ranking
    const int64_t waves = (n + 63) / 64;
    int parts = 1;
    while (parts < W6_MAX_PARTS && parts * 2 <= count && waves * (parts * 2 + 4) <= 1024 * BC6H_WIDE_WAVES) parts *= 2;
    const unsigned gx = (unsigned)((n + TPB6 - 1) / TPB6);
    const dim3 blk(TPB6);
    if (vec) hipLaunchKernelGGL((bc6h_wide_phaseA<true>),  dim3(gx, (unsigned)(parts + 4)), blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, s, parts, count);
    else     hipLaunchKernelGGL((bc6h_wide_phaseA<false>), dim3(gx, (unsigned)(parts + 4)), blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, s, parts, count);
    if (vec) hipLaunchKernelGGL((bc6h_wide_phaseB<true>),  dim3(gx, 6u), blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, s, parts);
    else     hipLaunchKernelGGL((bc6h_wide_phaseB<false>), dim3(gx, 6u), blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, s, parts);
    hipLaunchKernelGGL(bc6h_wide_commit, dim3(gx), blk, 0, st, cerr, cblk, (int32_t)n, 0x3ffu, reinterpret_cast<uint4*>(dst), vec ? 1 : 0);
}

void launch_bc6h(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst,
                 const bc6h_enc_settings& s, hipStream_t st, void* workspace)
{
    const int bx = width / 4, by = height / 4;
    const int64_t n = (int64_t)bx * by;
    if (n <= 0) return;
    if (workspace && bc6h_use_wide(n, s)) { launch_bc6h_wide(src, stride, bx, n, dst, s, workspace, st); return; }
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const dim3 grid((unsigned)((n + TPB6 - 1) / TPB6)), blk(TPB6);
    if (s.slow_mode) {
        if (vec) hipLaunchKernelGGL((bc6h_kernel<true, true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst, s);
        else     hipLaunchKernelGGL((bc6h_kernel<true, false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst, s);
    } else if (s.fastSkipTreshold <= 0) {
        if (vec) hipLaunchKernelGGL((bc6h_one_region_kernel<true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst, s);
        else     hipLaunchKernelGGL((bc6h_one_region_kernel<false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst, s);
    } else {
        if (vec) hipLaunchKernelGGL((bc6h_kernel<false, true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst, s);
        else     hipLaunchKernelGGL((bc6h_kernel<false, false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst, s);
    }
}

} // namespace itw
