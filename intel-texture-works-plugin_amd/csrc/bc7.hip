// bc7.hip -- BC7 encoder kernels for gfx950 (MI355X).
//
// Replaces kernel.ispc:616-2037 (CompressBlocksBC7_ispc) behind CompressBlocksBC7
// (ispc_texcomp.cpp:427-430).  Same search as the reference -- PCA line fits per
// subset, p-bit aware endpoint quantisation, two-candidate index selection,
// least-squares refinement of each mode's winner, modes 0-7 -- organised for a
// SIMT machine instead of an SPMD gang:
//
//   * one 4x4 block per lane, read from HBM with 4 coalesced dwordx4 loads per lane,
//     16 B written per lane.  The block lives in 16 VGPRs as loaded plus 16 VGPRs of
//     planar bytes (bc7_exact.hpp);
//   * everything the reference computes in fp32 that cannot round (texels and decoded
//     endpoints are integers <= 255) runs on the packed integer units: subset moment
//     sums are masked v_dot4_u32_u8 chains, index selection projects with v_dot2 and an
//     exactly rounded quotient and compares the two candidate indices through per-segment
//     palettes in LDS (one v_dot4 per candidate), least-squares sums are dot4 chains
//     (proofs in bc7_exact.hpp).  What can round (PCA, endpoint quantisation, the 2x2
//     solve) is fp32 with the pinned x86 arithmetic;
//   * the SCANS of the shapes (110-128 VGPRs, 4 waves per SIMD) and the REFINEMENT of each mode's winner (per-lane shapes,
//     2 waves per SIMD) have different register budgets and therefore live in different kernels; how many launches a call
//     takes depends on its size (bottom of this file): whole surfaces run FUSED (bc7_scan_all: the scans of families {0,2}
//     and {1,3} in one launch, XCD-aware so the second family reads its texels from L2; bc7_finish_all: every mode's
//     refinement + modes 4/5/6 in one launch), calls too small to fill the chip run WIDE (each scan split over several
//     waves, winners joined by an ordered argmin, single-subset modes on a second stream).  Families run in the
//     reference's order and only communicate through "best error so far" (kernel.ispc:1358, 1638, 1684) and the search
//     winners; the per-family search / finish kernel pairs of round 1 remain for ranked lists longer than 16 shapes.
//     The slow profiles (every shape scanned) run the BOUNDED order instead (round 4: modes 1/3/7 last, only for the blocks an
//     exact lower bound cannot exclude, compacted into lists) as two interleaved bands on two streams, and for the RGB
//     profile a pilot kernel picks that order or the reference's per call, on the device (round 5; launch_bc7);
//   * the scans produce a candidate's ERROR and nothing else: the level a texel takes
//     and the packed indices matter for a mode's winner alone, so the winner record is
//     {error, shape} and the finish kernel recomputes endpoints and indices of the
//     winner (same inputs, same bits) before refining it;
//   * shapes are visited in a wave-uniform order wherever the candidate list is the
//     whole table (modes 0/2 always; modes 1/3/7 when their fastSkipTreshold is >= 64,
//     the `slow` profiles): subset masks are scalars, a texel costs work only in the
//     subset it belongs to (scalar branches), and the loop over subsets is rolled so
//     every code path exists once.  The reference scans its PCA-ranked list with a
//     strict `<`, i.e. among equal errors the lowest rank key wins; here the rank key
//     (part + 64*bound) is only evaluated when two shapes actually tie;
//   * modes 0/2: a subset's mode 2 result depends on its texel mask alone and 192
//     subsets use 140 masks, so the scan follows a generated schedule that clusters
//     equal masks and reloads results from a four-entry register cache;
//   * shorter ranked lists (fast profiles) keep the per-lane order: the i-th entry of
//     the reference's selection sort (kernel.ispc:1365-1384) is the i-th smallest key --
//     the 16 smallest are kept sorted in registers while the keys are produced (every
//     preset; longer custom lists scan 64 keys in LDS), no dynamic register indexing;
//   * fits that do not depend on the mode are shared: shapes 64..79 serve modes 0 and
//     2, every two-subset shape serves modes 1 and 3 (kernel.ispc:1286-1291), the last
//     subset's moments are the block's minus the others' (exact), one rotation's fit
//     serves its three mode 4/5 candidates;
//   * the RCPPS/RSQRTPS seed tables are staged in LDS once per workgroup.
//
// VALU issue bound; nothing GEMM shaped, so no MFMA.  Bit-exactness with
// the oracle forbids FMA contraction and any re-association of sums that can round.
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include "bc7_exact.hpp"
#include "kernels.hpp"
#include "host_rt.hpp"

namespace itw {

#define BCN_TABLE_QUAL __device__ const
#include "bc7_f02_schedule.h"
#undef BCN_TABLE_QUAL

#ifndef ITW_BC7_LANE_PAL
#define ITW_BC7_LANE_PAL 1                // per-lane shapes (refinement, ranked lists <= 16) through per-subset palettes in LDS (bc7_exact.hpp)
#endif
constexpr int LANE_PAL_LEVELS = 24;       // 3 subsets x 8 levels: the most one mode's refinement decodes (48 KiB per workgroup)
constexpr int TPB = 256;                  // four waves share one staged seed table
constexpr float INV255 = 1.0f / 255.0f;   // x/255f under fast-math = x*(1.f/255.f)
constexpr int32_t ERR_MAX = 0x7fffffff;
enum Family { F_MODES02 = 0, F_MODES13 = 1, F_MODE7 = 2, F_MODES456 = 3 };

struct ModeTraits { int pairs, bits, ch; };
__device__ __forceinline__ constexpr ModeTraits traits(int mode)
{
    return mode == 0 ? ModeTraits{3, 3, 3} : mode == 1 ? ModeTraits{2, 3, 3} : mode == 2 ? ModeTraits{3, 2, 3}
         : mode == 3 ? ModeTraits{2, 2, 3} : ModeTraits{2, 2, 4} /* 7 */;
}

__device__ __forceinline__ int32_t expand_to_byte(int32_t v, int bits)        // kernel.ispc:976-981
{
    const int32_t vv = v << (8 - bits);
    return vv + (int32_t)((uint32_t)vv >> bits);
}

// float -> int of the reference (cvttps2dq).  SAFE: the caller guarantees a finite value inside the int range
// (search-time endpoints are clamped to [0,255] by the fit), so the plain conversion gives the same result.
template <bool SAFE>
__device__ __forceinline__ int32_t f2i(float f) { return SAFE ? (int32_t)f : f2i_x86(f); }

// ---- endpoint quantisers --------------------------------------------------------------------------
// in : e[i][p]  fitted endpoint i, channel p (fp32)
// out: q[i][p]  code as stored in the block (incl. p-bit in bit 0 where the mode has one)
//      d[i][p]  what a decoder reconstructs from it (the reference overwrites e with this; kernel.ispc:1100-1128)

// modes 0,3,6,7: one p-bit per endpoint, chosen by squared error over `err_ch` channels.  [kernel.ispc:983-1022]
// SAFE (fitted endpoints, clamped to [0,255]): u = (e/255*L2 - b)/2 + 0.5 is in [0, 2^BITS + 1), so the code
// 2*floor(u) + b stays an integer-valued float end to end (floor, *2, +b are exact) and only the chosen hypothesis is
// converted; of the reference's clamp [b, L2-1+b] only the upper bound of b = 0 can fire (u >= 2^BITS needs
// e/255*L2 >= L2, i.e. e = 255; for b = 1, u < 2^BITS always).  Modes 3, 6, 7 compare the code itself (kernel.ispc:
// 1003-1017: mode 7 compares raw 6-bit codes against 8-bit targets, a reference quirk), mode 0 its 5-bit expansion.
// For b = 1 the reference's u = (t - 1)*0.5 + 0.5 equals t*0.5 exactly when t >= 0.5 (t - 1 is then exact, the rest
// is scaling and an exactly representable sum); for t < 0.5 both floors are 0.  SAFE paths use floor(t*0.5) directly.
// Round 4: mode 0 (4 bits + p-bit, compared through its 5-bit expansion) and mode 1 (shared p-bit, 7-bit expansion) take the same
// float-domain route in the scans: a code c is an integer-valued float, and expand_to_byte(c, BITS) = c * 2^(8-BITS) + (c >> (2*BITS-8))
// is c * 8 + floor(c / 4) for 5 bits and c * 2 + [c >= 64] for 7 bits -- exact in fp32 (small integers, power-of-two scalings) -- so
// the hypotheses' decoded values are formed without a float -> int -> float round trip per value (the integer route costs a
// conversion each way plus a shift, a bit-field extract and an add, all 4-cycle forms; tests/test_exact_forms.py walks every code).
template <int MODE, bool SAFE>
__device__ __forceinline__ void quant_pbit(int32_t (&q)[2][4], int32_t (&d)[2][4], const float (&e)[2][4], int err_ch)
{
    constexpr int BITS = (MODE == 0) ? 4 : (MODE == 7) ? 5 : 7;
    constexpr int L2 = (1 << BITS) * 2 - 1;
    #pragma unroll
    for (int i = 0; i < 2; i++) {
        if (SAFE) {
            float cb[2][4], db[2][4];          // code incl. p-bit / what the error is measured against, per hypothesis
            #pragma unroll
            for (int p = 0; p < 4; p++) {
                const float t = e[i][p] * INV255 * (float)L2;
                const float v0 = __builtin_floorf(t * 0.5f + 0.5f) * 2.0f;
                cb[0][p] = __builtin_fminf(v0, (float)(L2 - 1));                   // v0 is an ordinary number here (SAFE)
                cb[1][p] = __builtin_floorf(t * 0.5f) * 2.0f + 1.0f;       // (t-1)/2 + 1/2 = t/2 exactly for t >= 0.5, floor 0 below
                #pragma unroll
                for (int b = 0; b < 2; b++)
                    db[b][p] = (MODE == 0) ? cb[b][p] * 8.0f + __builtin_floorf(cb[b][p] * 0.25f) : cb[b][p];   // mode 0: 5-bit code -> byte
            }
            float err0 = 0.f, err1 = 0.f;
            #pragma unroll
            for (int p = 0; p < 4; p++)
                if (p < err_ch) { err0 += sq(e[i][p] - db[0][p]); err1 += sq(e[i][p] - db[1][p]); }
            const bool first = err0 < err1;
            #pragma unroll
            for (int p = 0; p < 4; p++) {
                q[i][p] = (int32_t)(first ? cb[0][p] : cb[1][p]);
                d[i][p] = (MODE == 7) ? expand_to_byte(q[i][p], 6) : (MODE == 0) ? (int32_t)(first ? db[0][p] : db[1][p]) : q[i][p];
            }
            continue;
        }
        int32_t qb[2][4];
        float db[2][4];
        #pragma unroll
        for (int b = 0; b < 2; b++)
            #pragma unroll
            for (int p = 0; p < 4; p++) {
                const float t = e[i][p] * INV255 * (float)L2;
                const float u = (t - (float)b) * 0.5f + 0.5f;
                const uint32_t v = (uint32_t)f2i<SAFE>(u) * 2u + (uint32_t)b;
                qb[b][p] = iclamp((int32_t)v, b, L2 - 1 + b);
                // mode 0 compares in 8-bit space; modes 3/6 codes are 8-bit; mode 7 compares raw 6-bit codes
                // against 8-bit targets (reference quirk, kernel.ispc:1003-1017)
                db[b][p] = (float)((MODE == 0) ? expand_to_byte(qb[b][p], 5) : qb[b][p]);
            }
        float err0 = 0.f, err1 = 0.f;
        #pragma unroll
        for (int p = 0; p < 4; p++)
            if (p < err_ch) { err0 += sq(e[i][p] - db[0][p]); err1 += sq(e[i][p] - db[1][p]); }
        const bool first = err0 < err1;
        #pragma unroll
        for (int p = 0; p < 4; p++) {
            q[i][p] = first ? qb[0][p] : qb[1][p];
            d[i][p] = (MODE == 0) ? expand_to_byte(q[i][p], 5) : (MODE == 7) ? expand_to_byte(q[i][p], 6) : q[i][p];
        }
    }
}

// mode 1: one p-bit shared by both endpoints of a subset, RGB error.            [kernel.ispc:1024-1052]
template <bool SAFE>
__device__ __forceinline__ void quant_shared_pbit(int32_t (&q)[2][4], int32_t (&d)[2][4], const float (&e)[2][4])
{
    if (SAFE) {
        // float domain (see quant_pbit): code c in [0, 127], decoded byte c * 2 + (c >> 6) = c + c + [c >= 64]
        float cb[2][2][4], db[2][2][4];
        #pragma unroll
        for (int i = 0; i < 2; i++)
            #pragma unroll
            for (int p = 0; p < 4; p++) {
                const float t = e[i][p] * INV255 * 127.0f;
                const float v0 = __builtin_floorf(t * 0.5f + 0.5f) * 2.0f;
                cb[0][i][p] = __builtin_fminf(v0, 126.0f);
                cb[1][i][p] = __builtin_floorf(t * 0.5f) * 2.0f + 1.0f;
                #pragma unroll
                for (int b = 0; b < 2; b++)
                    db[b][i][p] = cb[b][i][p] * 2.0f + __builtin_floorf(cb[b][i][p] * 0.015625f);
            }
        float err0 = 0.f, err1 = 0.f;
        #pragma unroll
        for (int i = 0; i < 2; i++)
            #pragma unroll
            for (int p = 0; p < 3; p++) { err0 += sq(e[i][p] - db[0][i][p]); err1 += sq(e[i][p] - db[1][i][p]); }
        const bool first = err0 < err1;
        #pragma unroll
        for (int i = 0; i < 2; i++)
            #pragma unroll
            for (int p = 0; p < 4; p++) {
                q[i][p] = (int32_t)(first ? cb[0][i][p] : cb[1][i][p]);
                d[i][p] = (int32_t)(first ? db[0][i][p] : db[1][i][p]);
            }
        return;
    }
    int32_t qb[2][2][4];
    float db[2][2][4];
    #pragma unroll
    for (int b = 0; b < 2; b++)
        #pragma unroll
        for (int i = 0; i < 2; i++)
            #pragma unroll
            for (int p = 0; p < 4; p++) {
                const float t = e[i][p] * INV255 * 127.0f;
                const float u = (t - (float)b) * 0.5f + 0.5f;
                const uint32_t v = (uint32_t)f2i<SAFE>(u) * 2u + (uint32_t)b;
                qb[b][i][p] = iclamp((int32_t)v, b, 126 + b);
                db[b][i][p] = (float)expand_to_byte(qb[b][i][p], 7);
            }
    float err0 = 0.f, err1 = 0.f;
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int p = 0; p < 3; p++) { err0 += sq(e[i][p] - db[0][i][p]); err1 += sq(e[i][p] - db[1][i][p]); }
    const bool first = err0 < err1;
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int p = 0; p < 4; p++) {
            q[i][p] = first ? qb[0][i][p] : qb[1][i][p];
            d[i][p] = expand_to_byte(q[i][p], 7);
        }
}

// modes 2,4 (5 bits) and 5 (7 bits): plain rounding.                            [kernel.ispc:1054-1065]
template <int BITS, bool SAFE>
__device__ __forceinline__ void quant_plain(int32_t (&q)[2][4], int32_t (&d)[2][4], const float (&e)[2][4])
{
    constexpr int L = 1 << BITS;
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int p = 0; p < 4; p++) {
            const int32_t v = f2i<SAFE>(e[i][p] * INV255 * (float)(L - 1) + 0.5f);
            q[i][p] = SAFE ? v : iclamp(v, 0, L - 1);          // e in [0,255]: e/255*(L-1) + 0.5 < L, never negative
            d[i][p] = expand_to_byte(q[i][p], BITS);
        }
}

template <int MODE, bool SAFE>
__device__ __forceinline__ void quant_mode(int32_t (&q)[2][4], int32_t (&d)[2][4], const float (&e)[2][4], int err_ch)
{
    if (MODE == 0 || MODE == 3 || MODE == 6 || MODE == 7) quant_pbit<MODE, SAFE>(q, d, e, err_ch);
    else if (MODE == 1) quant_shared_pbit<SAFE>(q, d, e);
    else if (MODE == 2 || MODE == 4) quant_plain<5, SAFE>(q, d, e);
    else quant_plain<7, SAFE>(q, d, e);
}

// ---- per-lane state ---------------------------------------------------------------------------
// All BC7 block errors are exact integers below 2^24 in the reference's float arithmetic (sums of per-texel
// `(int)err`, plus the integer opaque_err), so they are carried as int32 here; +inf is ERR_MAX.
struct Lane {
    Tex tx;
    int32_t best_err;
    int32_t opaque_err;
    uint32_t best[4];
    bool improved;
    SeedTables T;
    int32_t* keys;            // LDS column (ranked-list path only): keys[i * TPB]
    uint2* pal;               // LDS column of palettes (table-order scans): 12 levels, pal[level * TPB]
};

struct Win {                  // winner of one multi-subset mode during the search (its indices are recomputed by the finish kernel)
    int32_t err;
    int32_t shape;            // table index 0..63 (two subsets) / 64..127 (three)
    int32_t key;              // rank key of `shape`; < 0 = not evaluated yet (table-order scans)
};

__device__ __forceinline__ void reset(Win& w, int shape0)
{
    w.err = ERR_MAX;
    w.shape = shape0;
    w.key = -1;
}

__device__ __forceinline__ void store_bits(uint32_t (&out)[4], const BlockBits& bb)
{
    out[0] = (uint32_t)bb.lo; out[1] = (uint32_t)(bb.lo >> 32); out[2] = (uint32_t)bb.hi; out[3] = (uint32_t)(bb.hi >> 32);
}

// ---- bitstream --------------------------------------------------------------------------------
// modes 0,1,2,3,7                                                        [kernel.ispc:1708-1733, 1767-1877]
template <int MODE>
__device__ __forceinline__ void emit_multi(uint32_t (&out)[4], int32_t (&cq)[3][2][4], const uint32_t (&cqb)[2], int shape)
{
    constexpr ModeTraits M = traits(MODE);
    constexpr int LEVELS = 1 << M.bits;
    const Shape sh = load_shape(shape);
    const int a1 = (int)(sh.anchors >> 4), a2 = (int)(sh.anchors & 15u);

    // an anchor index must have its top bit clear: swap the subset's endpoints and mirror its indices
    uint32_t flips = 0;
    #pragma unroll
    for (int j = 0; j < M.pairs; j++) {
        const int k0 = (j == 0) ? 0 : ((j == 1) ? a1 : a2);
        const uint32_t word = (k0 < 8) ? cqb[0] : cqb[1];
        const int32_t q = (int32_t)((word >> (4 * (k0 & 7))) & 15u);
        if (q >= LEVELS / 2) {
            #pragma unroll
            for (int p = 0; p < 4; p++) { const int32_t t = cq[j][0][p]; cq[j][0][p] = cq[j][1][p]; cq[j][1][p] = t; }
            flips |= subset_mask(sh, j);
        }
    }

    BlockBits bb;
    int pos = 0;
    bb.put(pos, MODE + 1, 1u << MODE); pos += MODE + 1;
    if (MODE == 0) { bb.put(pos, 4, (uint32_t)shape & 15u); pos += 4; }
    else           { bb.put(pos, 6, (uint32_t)shape & 63u); pos += 6; }

    constexpr int EPB = (MODE == 0) ? 4 : (MODE == 1) ? 6 : (MODE == 2) ? 5 : (MODE == 3) ? 7 : 5;
    constexpr bool HAS_P = (MODE != 2);
    #pragma unroll
    for (int p = 0; p < M.ch; p++)
        #pragma unroll
        for (int j = 0; j < M.pairs; j++)
            #pragma unroll
            for (int i = 0; i < 2; i++) {
                bb.put(pos, EPB, (uint32_t)(HAS_P ? (cq[j][i][p] >> 1) : cq[j][i][p]));
                pos += EPB;
            }
    if (MODE == 1) {
        #pragma unroll
        for (int j = 0; j < 2; j++) { bb.put(pos, 1, (uint32_t)cq[j][0][0] & 1u); pos += 1; }
    } else if (HAS_P) {
        #pragma unroll
        for (int j = 0; j < M.pairs; j++)
            #pragma unroll
            for (int i = 0; i < 2; i++) { bb.put(pos, 1, (uint32_t)cq[j][i][0] & 1u); pos += 1; }
    }

    const int start = pos;                                   // = 128 + (pairs-1) - (16*bits - 1)
    #pragma unroll
    for (int k = 0; k < 16; k++) {
        uint32_t q = ((k < 8 ? cqb[0] >> (4 * k) : cqb[1] >> (4 * (k - 8))) & 15u);
        if ((flips >> k) & 1u) q = (uint32_t)(LEVELS - 1) - q;
        const int n = (k == 0) ? M.bits - 1 : M.bits;
        bb.put(pos, n, q); pos += n;
    }
    // delete the (zero) top bit of the other anchors, highest position first
    const int msb1 = start + M.bits * a1 + M.bits - 2;
    if (M.pairs == 3) {
        const int msb2 = start + M.bits * a2 + M.bits - 2;
        bb.drop_bit(max(msb1, msb2));
        bb.drop_bit(min(msb1, msb2));
    } else {
        bb.drop_bit(msb1);
    }
    store_bits(out, bb);
}

// single-subset anchor rule for modes 4,5,6                                       [kernel.ispc:1694-1706]
template <int BITS, int NCH>
__device__ __forceinline__ void fix_anchor(int32_t (&e0)[NCH], int32_t (&e1)[NCH], uint32_t (&qb)[2])
{
    constexpr uint32_t L = 1u << BITS;
    if ((qb[0] & 15u) >= L / 2) {
        #pragma unroll
        for (int p = 0; p < NCH; p++) { const int32_t t = e0[p]; e0[p] = e1[p]; e1[p] = t; }
        qb[0] = 0x11111111u * (L - 1) - qb[0];
        qb[1] = 0x11111111u * (L - 1) - qb[1];
    }
}

template <int BITS>
__device__ __forceinline__ void put_indices(BlockBits& bb, int& pos, const uint32_t (&qb)[2])
{
    #pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t q = ((k < 8 ? qb[0] >> (4 * k) : qb[1] >> (4 * (k - 8))) & 15u);
        const int n = (k == 0) ? BITS - 1 : BITS;
        bb.put(pos, n, q); pos += n;
    }
}

struct Dual {                 // a mode 4/5 candidate: vector part + scalar part
    int32_t q[2][4];          // colour endpoints (slot 3 unused)
    uint32_t qb[2];
    int32_t aq[2];            // scalar channel endpoints
    uint32_t aqb[2];
    int32_t rotation, swap;
};

// modes 4 and 5                                                                   [kernel.ispc:1879-1939]
template <int MODE>
__device__ __forceinline__ void emit_dual(uint32_t (&out)[4], const Dual& d)
{
    constexpr int ABITS = (MODE == 4) ? 3 : 2, EPB = (MODE == 4) ? 5 : 7, AEPB = (MODE == 4) ? 6 : 8;
    int32_t c0[4], c1[4], a0[1], a1[1];
    uint32_t vb[2] = {d.qb[0], d.qb[1]}, sb[2] = {d.aqb[0], d.aqb[1]};
    #pragma unroll
    for (int p = 0; p < 4; p++) { c0[p] = d.q[0][p]; c1[p] = d.q[1][p]; }
    a0[0] = d.aq[0]; a1[0] = d.aq[1];
    if (!d.swap) {
        fix_anchor<2, 4>(c0, c1, vb);
        fix_anchor<ABITS, 1>(a0, a1, sb);
    } else {
        // index sets trade places: the 2-bit set (now the scalar's) is stored first
        uint32_t t0 = vb[0], t1 = vb[1]; vb[0] = sb[0]; vb[1] = sb[1]; sb[0] = t0; sb[1] = t1;
        fix_anchor<2, 1>(a0, a1, vb);
        fix_anchor<ABITS, 4>(c0, c1, sb);
    }
    BlockBits bb;
    int pos = 0;
    bb.put(pos, MODE + 1, 1u << MODE); pos += MODE + 1;
    bb.put(pos, 2, (uint32_t)(d.rotation + 1) & 3u); pos += 2;
    if (MODE == 4) { bb.put(pos, 1, (uint32_t)d.swap); pos += 1; }
    #pragma unroll
    for (int p = 0; p < 3; p++) {
        bb.put(pos, EPB, (uint32_t)c0[p]); pos += EPB;
        bb.put(pos, EPB, (uint32_t)c1[p]); pos += EPB;
    }
    bb.put(pos, AEPB, (uint32_t)a0[0]); pos += AEPB;
    bb.put(pos, AEPB, (uint32_t)a1[0]); pos += AEPB;
    put_indices<2>(bb, pos, vb);
    put_indices<ABITS>(bb, pos, sb);
    store_bits(out, bb);
}

// mode 6                                                                          [kernel.ispc:1941-1964]
__device__ __forceinline__ void emit_mode6(uint32_t (&out)[4], int32_t (&q)[2][4], uint32_t (&qb)[2])
{
    fix_anchor<4, 4>(q[0], q[1], qb);
    BlockBits bb;
    int pos = 0;
    bb.put(pos, 7, 64u); pos += 7;
    #pragma unroll
    for (int p = 0; p < 4; p++) {
        bb.put(pos, 7, (uint32_t)(q[0][p] >> 1)); pos += 7;
        bb.put(pos, 7, (uint32_t)(q[1][p] >> 1)); pos += 7;
    }
    bb.put(pos, 1, (uint32_t)q[0][0] & 1u); pos += 1;
    bb.put(pos, 1, (uint32_t)q[1][0] & 1u); pos += 1;
    put_indices<4>(bb, pos, qb);
    store_bits(out, bb);
}

// ---- multi-subset modes ---------------------------------------------------------------------

// Rank key of a two-subset shape: shape + 64 * (int)(sqrt(residual bound) * 256).   [kernel.ispc:952-971, 1404-1409]
// `shape` may differ per lane (table loads are then per lane).
template <int RANK_CH>
__device__ __forceinline__ int32_t rank_key(int shape, const Tex& tx, const Stats<RANK_CH>& full, const SeedTables& T)
{
    const SubsetMask sm = subset_of(shape, 0);
    IStats<RANK_CH> s0;
    stats_int<RANK_CH>(s0, tx.pl, sm);
    Stats<RANK_CH> f0;
    stats_float<RANK_CH>(f0, s0);
    return (int32_t)((uint32_t)shape + (uint32_t)split_bound_from<RANK_CH, true>(f0, full, T, rcp_of_count(sm.n), rcp_of_count(16 - sm.n)) * 64u);
}

// Least-squares refinement of a mode's winner, then the mode competes for the block.  [kernel.ispc:1329-1362]
// The winner's shape differs per lane: masks come from per-lane table loads, segments are chosen per texel.
template <int MODE>
__device__ __forceinline__ void refine_and_commit(Lane& ln, Win& w, int iterations, int settings_channels)
{
    constexpr ModeTraits M = traits(MODE);
    const Shape sh = load_shape(w.shape);
    SubsetMask sm[3];
    #pragma unroll
    for (int j = 0; j < M.pairs; j++) sm[j] = subset_of(w.shape, j);
    // endpoint codes and indices of the search-time winner: the scan kept only its error, so refit + quantise its
    // shape again (same inputs, same bits) and select once; the error found equals w.err
    int32_t cq[3][2][4];
    uint32_t wqb[2];
    #pragma unroll
    for (int j = 0; j < 3; j++) for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) cq[j][i][p] = 0;
#if ITW_BC7_LANE_PAL
    constexpr int LEVELS = 1 << M.bits;
    const int32_t tt = block_norm2<M.ch>(ln.tx.pl);
#endif
    {
#if ITW_BC7_LANE_PAL
        PalSegment sg0[3];
#else
        Segment sg0[3];
#endif
        #pragma unroll 1
        for (int j = 0; j < M.pairs; j++) {
            const SubsetMask s = subset_of(w.shape, j);       // (selecting among sm[0..2] here made the compiler index them in scratch)
            IStats<M.ch> st;
            stats_int<M.ch>(st, ln.tx.pl, s);
            float ep[2][4];
            ep[0][3] = 0.f; ep[1][3] = 0.f;
            fit_line<M.ch>(ep, ln.tx, s.bits, st, ispc_rcp((float)s.n, ln.T), ln.T);
            int32_t q[2][4], d[2][4];
            quant_mode<MODE, true>(q, d, ep, M.ch);
#if ITW_BC7_LANE_PAL
            const PalSegment sgj = build_palette<M.bits, M.ch, TPB>(ln.pal + j * (LEVELS * TPB), d);
#else
            const Segment sgj = make_segment<M.bits, M.ch>(d);
#endif
            #pragma unroll
            for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) {
                if (j == 0) cq[0][i][p] = q[i][p]; else if (j == 1) cq[1][i][p] = q[i][p]; else cq[2][i][p] = q[i][p];
            }
            if (j == 0) sg0[0] = sgj; else if (j == 1) sg0[1] = sgj; else sg0[2] = sgj;
        }
        if (M.pairs == 2) sg0[2] = sg0[1];
#if ITW_BC7_LANE_PAL
        (void)select_block_lanes_pal<M.bits, M.ch, M.pairs, TPB, true>(wqb, ln.tx, sg0, ln.pal, sh.pattern, tt);
#else
        (void)select_block<M.bits, M.ch, M.pairs>(wqb, ln.tx, sg0, sh.pattern);
#endif
    }
    for (int it = 0; it < iterations; it++) {
        ln.tx.fence();
        int32_t q[3][2][4];
#if ITW_BC7_LANE_PAL
        PalSegment sg[3];
#else
        Segment sg[3];
#endif
        #pragma unroll
        for (int j = 0; j < M.pairs; j++) {
            float ep[2][4];
            int32_t d[2][4];
            ep[0][3] = 0.f; ep[1][3] = 0.f;
            refit_line<M.bits, M.ch>(ep, ln.tx.pl, wqb, sm[j], ln.T);
            quant_mode<MODE, false>(q[j], d, ep, settings_channels);       // :1343 passes the profile's channel count
#if ITW_BC7_LANE_PAL
            sg[j] = build_palette<M.bits, M.ch, TPB>(ln.pal + j * (LEVELS * TPB), d);
#else
            sg[j] = make_segment<M.bits, M.ch>(d);
#endif
        }
        if (M.pairs == 2) sg[2] = sg[1];
        uint32_t qb[2];
#if ITW_BC7_LANE_PAL
        const int32_t err = select_block_lanes_pal<M.bits, M.ch, M.pairs, TPB, true>(qb, ln.tx, sg, ln.pal, sh.pattern, tt);
#else
        const int32_t err = select_block<M.bits, M.ch, M.pairs>(qb, ln.tx, sg, sh.pattern);
#endif
        const bool better = err < w.err;
        if (better) {
            #pragma unroll
            for (int j = 0; j < M.pairs; j++) for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) cq[j][i][p] = q[j][i][p];
            wqb[0] = qb[0]; wqb[1] = qb[1];
            w.err = err;
        }
        // an iteration is a function of the kept indices alone (kernel.ispc:1331-1356): one that improves nothing leaves them
        // as they were, so every later one repeats it.  Once that holds for all 64 blocks of the wave the rest is skipped.
        if (__all(!better)) break;
    }
    int32_t err = w.err;
    if (MODE != 7) err += ln.opaque_err;
    if (err < ln.best_err) {
        ln.best_err = err;
        ln.improved = true;
        emit_multi<MODE>(ln.best, cq, wqb, w.shape);
    }
}

__device__ __forceinline__ void take(Win& w, int32_t err, int shape, int32_t key)
{
    w.err = err; w.shape = shape; w.key = key;
}

// modes 0 and 2: three subsets, every shape a candidate, one fit per subset.              [kernel.ispc:1386-1394]
// The reference scans the shapes in table order with a strict `<`: the winner is the lowest error and, among equal
// errors, the lowest table index -- whatever the visiting order.  With mode 2 enabled the scan therefore follows
// BC7_F02_SCHEDULE (tools/gen_bc7_f02_schedule.py): 192 (shape, subset) pairs use only 140 distinct texel masks and a
// subset's whole mode 2 result (indices, error) depends on the mask alone, so the order clusters equal masks and a
// four-entry register cache, driven by the wave-uniform schedule word, turns 39 subset evaluations into loads (9 more
// keep their fit for mode 0 and skip the mode 2 part).
// A cached subset result is its error (the scans carry no indices).
// The cache is four scalars (an indexed aggregate would be demoted to scratch); `slot` is wave-uniform.
#define ITW_CACHE_GET(F, slot) ((slot) == 0u ? F##0 : ((slot) == 1u ? F##1 : ((slot) == 2u ? F##2 : F##3)))
#define ITW_CACHE_PUT(F, slot, v) do { F##0 = (slot) == 0u ? (v) : F##0; F##1 = (slot) == 1u ? (v) : F##1; F##2 = (slot) == 2u ? (v) : F##2; F##3 = (slot) == 3u ? (v) : F##3; } while (0)

// `first`/`step`: this call visits positions first, first + step, ... (the wide path splits one scan over several waves,
// WIDE below); a split scan takes the shapes in table order without the cache (any visiting order gives the same winner).
__device__ __forceinline__ void search_02(Lane& ln, const bc7_enc_settings& S, Win& b0, Win& b2, int first = 0, int step = 1)
{
    reset(b0, 64); reset(b2, 64);
    IStats<3> full;
    stats_int<3>(full, ln.tx.pl, whole_block());
    const int32_t tt = full.m[0] + full.m[4] + full.m[7];          // sum over the block of |texel|^2
    const bool do2 = !S.skip_mode2;
    const int count = do2 ? 64 : 16;
    int32_t ce0 = 0, ce1 = 0, ce2 = 0, ce3 = 0;      // cached subset results (mode 2 error of a texel mask)
    for (int pos = first; pos < count; pos += step) {
        ln.tx.fence();
        const uint32_t sched = (do2 && step == 1) ? BC7_F02_SCHEDULE[pos] : (uint32_t)pos;     // mode 0 alone / split scan: table order, nothing cached
        const int part = (int)(sched & 63u);
        const bool do0 = part < 16;
        int32_t e0 = 0, e2 = 0;
        IStats<3> rest = full;
        bool rest_valid = true;                          // `rest` = block minus the subsets handled so far
        #pragma unroll 1
        for (int j = 0; j < 3; j++) {
            ln.tx.fence();                                // texel-derived values are used once per shape: do not hoist
            const uint32_t act = (sched >> (8 + 4 * j)) & 3u, slot = (sched >> (10 + 4 * j)) & 3u;
            const bool load = act == 2u;
            if (load) {                                   // this mask's mode 2 result is in the cache
                e2 += ITW_CACHE_GET(ce, slot);
                if (!do0) { rest_valid = false; continue; }
            }
            const SubsetMask sm = subset_of(64 + part, j);
            IStats<3> st;
            if (j < 2 || !rest_valid) stats_int<3>(st, ln.tx.pl, sm); else st = rest;
            stats_sub<3>(rest, st);
            float fit[2][4];
            fit[0][3] = 0.f; fit[1][3] = 0.f;            // the reference's unwritten alpha slots, pinned to 0
            fit_line<3>(fit, ln.tx, sm.bits, st, rcp_of_count(sm.n), ln.T);
            int32_t q[2][4], d[2][4];
            if (do0) {
                quant_mode<0, true>(q, d, fit, 3);
                const PalSegment ps = build_palette<3, 3, TPB>(ln.pal, d);
                subset_error_pal<3, 3, TPB>(e0, ln.tx, ps, ln.pal, sm.bits);
            }
            if (do2 && !load) {
                quant_mode<2, true>(q, d, fit, 3);
                const PalSegment ps = build_palette<2, 3, TPB>(ln.pal + 8 * TPB, d);
                int32_t r = 0;
                subset_error_pal<2, 3, TPB>(r, ln.tx, ps, ln.pal + 8 * TPB, sm.bits);
                e2 += r;
                if (act == 1u) ITW_CACHE_PUT(ce, slot, r);
            }
        }
        e0 += tt; e2 += tt;                              // the |t|^2 terms the palette path leaves out
        const int shape = 64 + part;
        if (do0 && (e0 < b0.err || (e0 == b0.err && shape < b0.shape))) take(b0, e0, shape, part);
        if (do2 && (e2 < b2.err || (e2 == b2.err && shape < b2.shape))) take(b2, e2, shape, part);
    }
}

// Two-subset modes.  FAMILY7 = false: modes 1 and 3 (3-channel fit, shared);  true: mode 7 (4-channel fit).
// RANK_CH: channels used by the PCA ranking (3 for modes 1/3; the profile's channel count for mode 7).
// RANKED: 0 = every shape is a candidate (table-order scan); 1 = at least one of the family's lists is a proper prefix
// of the PCA ranking (fast profiles), keys in LDS; 2 = the same with lists of at most 16 shapes (every preset of the
// reference): the 16 smallest keys are kept sorted in registers while the keys are produced, no LDS.
// `first`/`step`: the share of a split scan (wide path): table-order scans visit shapes first, first + step, ...; ranked
// scans compute all 64 keys and evaluate list entries first, first + step, ...
template <bool FAMILY7, int RANK_CH, int RANKED>
__device__ __forceinline__ void search_two_subset(Lane& ln, const bc7_enc_settings& S, Win& wa, Win& wb, int first = 0, int step = 1)
{
    constexpr int FIT_CH = FAMILY7 ? 4 : 3;
    const int na = FAMILY7 ? S.fastSkipTreshold_mode7 : S.fastSkipTreshold_mode1;   // first mode of the family
    const int nb = FAMILY7 ? 0 : S.fastSkipTreshold_mode3;                           // second mode
    reset(wa, 0); reset(wb, 0);
    if (na <= 0 && nb <= 0) return;

    IStats<FIT_CH> full;
    stats_int<FIT_CH>(full, ln.tx.pl, whole_block());
    const int32_t tt = full.m[0] + full.m[4] + full.m[7] + (FIT_CH == 4 ? full.m[9] : 0);     // sum over the block of |texel|^2
    Stats<RANK_CH> rfull;                                 // the ranking's view of the block (first RANK_CH channels)
    {
        IStats<RANK_CH> t;
        #pragma unroll
        for (int i = 0; i < 10; i++) t.m[i] = full.m[i];
        #pragma unroll
        for (int i = 0; i < 4; i++) t.s[i] = full.s[i];
        t.n = 16;
        stats_float<RANK_CH>(rfull, t);
    }

    if (RANKED == 0) {
        // every shape is a candidate: table order; the rank key is only needed to order shapes of equal error
        for (int part = first; part < 64; part += step) {
            ln.tx.fence();
            int32_t ea = 0, ec = 0;
            IStats<FIT_CH> rest = full;
            #pragma unroll 1
            for (int j = 0; j < 2; j++) {
                ln.tx.fence();                            // texel-derived values are used once per shape: do not hoist
                const SubsetMask sm = subset_of(part, j);
                IStats<FIT_CH> st;
                if (j == 0) stats_int<FIT_CH>(st, ln.tx.pl, sm); else st = rest;
                stats_sub<FIT_CH>(rest, st);
                float fit[2][4];
                fit[0][3] = 0.f; fit[1][3] = 0.f;
                fit_line<FIT_CH>(fit, ln.tx, sm.bits, st, rcp_of_count(sm.n), ln.T);
                int32_t q[2][4], d[2][4];
                if (FAMILY7) {
                    quant_mode<7, true>(q, d, fit, 4);
                    const PalSegment ps = build_palette<2, 4, TPB>(ln.pal, d);
                    subset_error_pal<2, 4, TPB>(ea, ln.tx, ps, ln.pal, sm.bits);
                } else {
                    if (na > 0 && nb > 0) {
                        quant_mode<1, true>(q, d, fit, 3);
                        const PalSegment s1 = build_palette<3, 3, TPB>(ln.pal, d);
                        quant_mode<3, true>(q, d, fit, 3);
                        const PalSegment s3 = build_palette<2, 3, TPB>(ln.pal + 8 * TPB, d);
                        subset_error2_pal<3, 2, 3, TPB>(ea, ec, ln.tx, s1, ln.pal, s3, ln.pal + 8 * TPB, sm.bits);
                    } else if (na > 0) {
                        quant_mode<1, true>(q, d, fit, 3);
                        const PalSegment ps = build_palette<3, 3, TPB>(ln.pal, d);
                        subset_error_pal<3, 3, TPB>(ea, ln.tx, ps, ln.pal, sm.bits);
                    } else {
                        quant_mode<3, true>(q, d, fit, 3);
                        const PalSegment ps = build_palette<2, 3, TPB>(ln.pal + 8 * TPB, d);
                        subset_error_pal<2, 3, TPB>(ec, ln.tx, ps, ln.pal + 8 * TPB, sm.bits);
                    }
                }
            }
            ea += tt; ec += tt;                          // the |t|^2 terms the palette path leaves out
            const bool on_a = na > 0, on_b = !FAMILY7 && nb > 0;
            const bool tie_a = on_a && ea == wa.err, tie_b = on_b && ec == wb.err;
            if (on_a && ea < wa.err) take(wa, ea, part, -1);
            if (on_b && ec < wb.err) take(wb, ec, part, -1);
            if (tie_a || tie_b) {
                // equal errors: the reference keeps whichever comes first in its ranked list = the lower key.
                // One rolled loop evaluates the missing keys: this shape's, then the incumbents'.
                int32_t key_here = 0;
                #pragma unroll 1
                for (int which = 0; which < 3; which++) {
                    const int shape = (which == 0) ? part : ((which == 1) ? wa.shape : wb.shape);
                    const bool need = (which == 0) || (which == 1 && tie_a && wa.key < 0) || (which == 2 && tie_b && wb.key < 0);
                    if (need) {
                        const int32_t k = rank_key<RANK_CH>(shape, ln.tx, rfull, ln.T);
                        if (which == 0) key_here = k; else if (which == 1) wa.key = k; else wb.key = k;
                    }
                }
                if (tie_a && key_here < wa.key) take(wa, ea, part, key_here);
                if (tie_b && key_here < wb.key) take(wb, ec, part, key_here);
            }
        }
    } else {
        // ranked prefix: the i-th list entry of the reference's selection sort is the i-th smallest key (keys are
        // distinct: their low 6 bits are the shape)                                      [kernel.ispc:1400-1414]
        int32_t top[16];
        #pragma unroll
        for (int i = 0; i < 16; i++) top[i] = 0x7fffffff;
        const int nlist = min(max(na, nb), 16);                   // entries of the ranked list that will be read (wave-uniform)
        for (int part = 0; part < 64; part++) {
            ln.tx.fence();
            const int32_t key = rank_key<RANK_CH>(part, ln.tx, rfull, ln.T);
            if (RANKED == 1) {
                ln.keys[part * TPB] = key;
            } else {
                // sorted insertion into the `nlist` smallest keys seen so far: a chain of compare-exchanges as long as the list the settings ask
                // for, in steps of four (wave-uniform: three scalar branches; `veryfast` keeps 3 keys, not 16: 24 of 32 VALU per shape saved)
                int32_t x = key;
#define ITW_INS4(b) { _Pragma("unroll") for (int i = (b); i < (b) + 4; i++) { const int32_t lo = min(top[i], x); x = max(top[i], x); top[i] = lo; } }
                ITW_INS4(0)
                if (nlist > 4) { ITW_INS4(4) if (nlist > 8) { ITW_INS4(8) if (nlist > 12) ITW_INS4(12) } }
#undef ITW_INS4
            }
        }
        const int n = min(max(na, nb), RANKED == 1 ? 64 : 16);
        int32_t prev = 0;
        for (int i = 0; i < n; i++) {
            if (RANKED == 1) {
                int32_t cur = 0x7fffffff;
                for (int t = 0; t < 64; t++) {
                    const int32_t k = ln.keys[t * TPB];
                    if ((i == 0 || k > prev) && k <= cur) cur = k;
                }
                prev = cur;
            } else {
                prev = top[0];
                #pragma unroll
                for (int t = 0; t < 15; t++) top[t] = top[t + 1];
            }
            if (step > 1 && (i % step) != first) continue;       // another wave's share of the list
            ln.tx.fence();
            const int shape = prev & 63;
            const Shape sh = load_shape(shape);
#if ITW_BC7_LANE_PAL
            if (RANKED == 2) {
                // lists of at most 16 shapes (every preset): the candidate's subsets decode their palettes into the lane's LDS column
                // (mode 1: 2 x 8 levels; mode 3 afterwards over the same stretch: 2 x 4) and the texels are scored through them
                float fits[2][2][4];
                IStats<FIT_CH> rest2 = full;
                #pragma unroll
                for (int j = 0; j < 2; j++) {
                    const SubsetMask sm = subset_of(shape, j);
                    IStats<FIT_CH> st;
                    if (j == 0) stats_int<FIT_CH>(st, ln.tx.pl, sm); else st = rest2;
                    stats_sub<FIT_CH>(rest2, st);
                    fits[j][0][3] = 0.f; fits[j][1][3] = 0.f;
                    fit_line<FIT_CH>(fits[j], ln.tx, sm.bits, st, ispc_rcp((float)sm.n, ln.T), ln.T);
                }
                uint32_t qbl[2];
                int32_t q[2][4], d[2][4];
                PalSegment ps[3];
                if (FAMILY7) {
                    #pragma unroll
                    for (int j = 0; j < 2; j++) { quant_mode<7, true>(q, d, fits[j], 4); ps[j] = build_palette<2, 4, TPB>(ln.pal + j * (4 * TPB), d); }
                    ps[2] = ps[1];
                    const int32_t e = select_block_lanes_pal<2, 4, 2, TPB, false>(qbl, ln.tx, ps, ln.pal, sh.pattern, tt);
                    if (e < wa.err) take(wa, e, shape, prev);
                } else {
                    if (i < na) {
                        #pragma unroll
                        for (int j = 0; j < 2; j++) { quant_mode<1, true>(q, d, fits[j], 3); ps[j] = build_palette<3, 3, TPB>(ln.pal + j * (8 * TPB), d); }
                        ps[2] = ps[1];
                        const int32_t e = select_block_lanes_pal<3, 3, 2, TPB, false>(qbl, ln.tx, ps, ln.pal, sh.pattern, tt);
                        if (e < wa.err) take(wa, e, shape, prev);
                    }
                    if (i < nb) {
                        #pragma unroll
                        for (int j = 0; j < 2; j++) { quant_mode<3, true>(q, d, fits[j], 3); ps[j] = build_palette<2, 3, TPB>(ln.pal + j * (4 * TPB), d); }
                        ps[2] = ps[1];
                        const int32_t e = select_block_lanes_pal<2, 3, 2, TPB, false>(qbl, ln.tx, ps, ln.pal, sh.pattern, tt);
                        if (e < wb.err) take(wb, e, shape, prev);
                    }
                }
                continue;
            }
#endif
            Segment sa[3], sc[3];
            IStats<FIT_CH> rest = full;
            #pragma unroll
            for (int j = 0; j < 2; j++) {
                const SubsetMask sm = subset_of(shape, j);
                IStats<FIT_CH> st;
                if (j == 0) stats_int<FIT_CH>(st, ln.tx.pl, sm); else st = rest;
                stats_sub<FIT_CH>(rest, st);
                float fit[2][4];
                fit[0][3] = 0.f; fit[1][3] = 0.f;
                fit_line<FIT_CH>(fit, ln.tx, sm.bits, st, ispc_rcp((float)sm.n, ln.T), ln.T);
                int32_t q[2][4], d[2][4];
                if (FAMILY7) {
                    quant_mode<7, true>(q, d, fit, 4);
                    sa[j] = make_segment<2, 4>(d);
                } else {
                    quant_mode<1, true>(q, d, fit, 3);
                    sa[j] = make_segment<3, 3>(d);
                    quant_mode<3, true>(q, d, fit, 3);
                    sc[j] = make_segment<2, 3>(d);
                }
            }
            sa[2] = sa[1]; sc[2] = sc[1];
            uint32_t qb[2];
            if (FAMILY7) {
                const int32_t e = select_block<2, 4, 2>(qb, ln.tx, sa, sh.pattern);
                if (e < wa.err) take(wa, e, shape, prev);
            } else {
                if (i < na) {
                    const int32_t e = select_block<3, 3, 2>(qb, ln.tx, sa, sh.pattern);
                    if (e < wa.err) take(wa, e, shape, prev);
                }
                if (i < nb) {
                    const int32_t e = select_block<2, 3, 2>(qb, ln.tx, sc, sh.pattern);
                    if (e < wb.err) take(wb, e, shape, prev);
                }
            }
        }
    }
}

// ---- modes 4 and 5: vector part (3 channels) + one separately coded channel ------------------

template <int BITS>
__device__ __forceinline__ int32_t weight_single(int32_t q)                       // kernel.ispc:675-686
{
    if (BITS == 2) return (int32_t)__builtin_amdgcn_perm(0u, 0x402b1500u, (uint32_t)q | 0x0c0c0c00u);
    if (BITS == 3) return (int32_t)__builtin_amdgcn_perm(0x40372e25u, 0x1b120900u, (uint32_t)q | 0x0c0c0c00u);
    return (int32_t)(((uint32_t)q * 68u + 8u) >> 4);
}

// scalar channel: min/max endpoints, then `iters` rounds of LS refit.             [kernel.ispc:1437-1563]
// Values v are the block's bytes of channel `rot_ch` (16 integers); vp = the same channel in planar form.
// Integer-exact parts: decode (int)(((64-w)*e0 + w*e1 + 32)/64) = e0 + ((w*(e1-e0)+32) >> 6), squared errors,
// and all sums of channel_opt_endpoints (<= 16*15*255).  The projection keeps the reference's fp32 form:
// (v - e0) is exact, * rcp(e1 - e0 + 0.001f), * LEVELS (exact) + 0.5 (one FMA = two roundings here).
// The LEVELS decoded values of a round are written once to the lane's own palette column in LDS and a texel reads its two
// neighbouring levels: two weights, two multiplies and six integer adds/shifts per texel become one address and one read.
template <int BITS, int EPBITS>
__device__ __forceinline__ int32_t encode_scalar(uint32_t (&qb)[2], int32_t (&qe)[2], const Tex& tx, int rot_ch,
                                                 const uint32_t (&vp)[4], int iters, const SeedTables& T, uint2* lv)
{
    constexpr int LEVELS = 1 << BITS;
    constexpr uint32_t LM1 = LEVELS - 1;
    constexpr float L1 = (float)LM1;
    constexpr int EL = 1 << EPBITS;
    const uint32_t shift = 8u * (uint32_t)rot_ch;
    int32_t lo = 255, hi = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const int32_t v = (int32_t)((tx.w[k] >> shift) & 255u); lo = min(lo, v); hi = max(hi, v); }
    float ep[2] = {(float)lo, (float)hi};
    int32_t err = 0;
    uint32_t pqb0 = 0, pqb1 = 0;
    for (int round = 0; ; round++) {
        int32_t de[2];
        #pragma unroll
        for (int i = 0; i < 2; i++) {                                            // channel_quant_dequant
            qe[i] = iclamp((int32_t)(ep[i] * INV255 * (float)(EL - 1) + 0.5f), 0, EL - 1);   // ep in [0,255]
            de[i] = expand_to_byte(qe[i], EPBITS);
        }
        qb[0] = qb[1] = 0u;                                                      // channel_opt_quant
        err = 0;
        const int32_t span = de[1] - de[0];
        const float rspan = ispc_rcp((float)span + 0.001f, T);
        // entry l of the lane's own palette column holds the float bits of levels (l, l + 1): one 8-byte read per texel
        // (the column is the vector palette's, dead by now; same element type, so no type-based reordering against the
        // next candidate's palette stores; other lanes' entries belong to waves that may be anywhere)
        {
            float lvl[LEVELS];
#pragma unroll
            for (int l = 0; l < LEVELS; l++) {
                constexpr int D = LEVELS - 1;
                const int32_t w = (l * 128 + D) / (2 * D);                       // the format's weight of level l (compile time)
                lvl[l] = (float)((l == 0) ? de[0] : (l == D) ? de[1] : de[0] + ((w * span + 32) >> 6));
            }
#pragma unroll
            for (int l = 0; l + 1 < LEVELS; l++) lv[l * TPB] = make_uint2(__float_as_uint(lvl[l]), __float_as_uint(lvl[l + 1]));
        }
        // texel values, decoded levels and their differences are integers below 2^8, squares and their sum stay below
        // 2^24: the float forms below are exact, and fp32 mul / sub issue at twice the rate of the integer multiply
        const float de0f = (float)de[0];
        float errf = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            // planar words hold texels {0,2,4,6} {1,3,5,7} {8,10,12,14} {9,11,13,15}: one v_cvt_f32_ubyteN each
            const float vf = (float)((vp[(k >> 3) * 2 + (k & 1)] >> (8 * ((k & 7) >> 1))) & 255u);
            const float proj = (vf - de0f) * rspan;                              // (float)(v - de0) of the reference, exact
            const int32_t q1 = imed3((int32_t)__builtin_fmaf(proj, (float)LEVELS, 0.5f), 1, LEVELS - 1);
            const uint2 pr = lv[(q1 - 1) * TPB];
            const float x0 = __uint_as_float(pr.x) - vf, x1 = __uint_as_float(pr.y) - vf;
            const float e0 = x0 * x0, e1 = x1 * x1;
            const bool first = e0 < e1;
            const uint32_t q = (uint32_t)(first ? q1 - 1 : q1);
            if (k < 8) qb[0] |= q << (4 * k); else qb[1] |= q << (4 * (k - 8));
            errf += __builtin_fminf(e0, e1);
        }
        err = (int32_t)errf;
        if (round >= iters) break;
        // the next round's endpoints are a function of these indices (channel_opt_endpoints): if they repeat the previous
        // round's for every block of the wave, all further rounds repeat this one
        if (__all(round > 0 && qb[0] == pqb0 && qb[1] == pqb1)) break;
        pqb0 = qb[0]; pqb1 = qb[1];
        const uint32_t qn[4] = {qb[0] & 0x0f0f0f0fu, (qb[0] >> 4) & 0x0f0f0f0fu, qb[1] & 0x0f0f0f0fu, (qb[1] >> 4) & 0x0f0f0f0fu};
        uint32_t sq_ = 0, sqq = 0, ssum = 0, satb = 0;                           // channel_opt_endpoints
#pragma unroll
        for (int d = 0; d < 4; d++) {
            sq_ = udot4(qn[d], 0x01010101u, sq_);
            sqq = udot4(qn[d], qn[d], sqq);
            ssum = udot4(vp[d], 0x01010101u, ssum);
            satb = udot4(vp[d], LM1 * 0x01010101u - qn[d], satb);
        }
        const float sum_q = (float)sq_, sum_qq = (float)sqq, sum = (float)ssum, atb1 = (float)satb;
        const float atb2 = L1 * sum - atb1;
        const float cxx = 16.0f * (L1 * L1) - (2.0f * L1) * sum_q + sum_qq;
        const float cyy = sum_qq;
        const float cxy = L1 * sum_q - sum_qq;
        const float det = cxx * cyy - cxy * cxy;
        const float scale = L1 * ispc_rcp(det, T);
        ep[0] = fclamp_num((atb1 * cyy - atb2 * cxy) * scale, 0.f, 255.f);
        ep[1] = fclamp_num((atb2 * cxx - atb1 * cxy) * scale, 0.f, 255.f);
        if (fabsf(det) < 0.001f) { ep[0] = sum * 0.0625f; ep[1] = ep[0]; }
    }
    return err;
}

// one (mode, rotation, index-swap) candidate, given the rotation's line fit      [kernel.ispc:1565-1621]
// `bound`: an error this candidate must stay below to matter at all (<= best_err).  The candidate's error is its vector part's plus
// its scalar channel's (>= 0) and it is taken only on a strict `<`: where the vector part alone reaches the bound for all 64
// blocks of the wave, the scalar channel (up to refineIterations_channel + 1 rounds over the 16 texels) is not encoded.
template <int MODE, int SWAP>
__device__ __forceinline__ void try_dual(Dual& best, int32_t& best_err, int32_t bound, const Lane& ln, Tex& rot, const float (&fit)[2][4], int32_t tt,
                                         const bc7_enc_settings& S, int rotation)
{
    constexpr int BITS = SWAP ? 3 : 2;
    constexpr int ABITS = SWAP ? 2 : ((MODE == 4) ? 3 : 2);
    constexpr int AEPB = (MODE == 4) ? 6 : 8;
    const SubsetMask all = whole_block();
    rot.fence();                                    // values derived from the rotated texels must not stay live across candidates

    int32_t q[2][4], d[2][4];
    uint32_t qb[2];
    quant_mode<MODE, true>(q, d, fit, 3);
    PalSegment ps = build_palette<BITS, 3, TPB>(ln.pal, d);
    int32_t err = select_block_pal<BITS, 3, TPB>(qb, rot, ps, ln.pal, tt);
    const int iters = S.refineIterations[MODE];
    for (int it = 0; it < iters; it++) {
        float ep[2][4];
        ep[0][3] = 0.f; ep[1][3] = 0.f;
        const uint32_t was0 = qb[0], was1 = qb[1];
        refit_line<BITS, 3>(ep, rot.pl, qb, all, ln.T);
        quant_mode<MODE, false>(q, d, ep, 3);
        ps = build_palette<BITS, 3, TPB>(ln.pal, d);
        err = select_block_pal<BITS, 3, TPB>(qb, rot, ps, ln.pal, tt);
        // the iteration maps indices to (endpoints, indices, error) (kernel.ispc:1598-1603): unchanged indices are a fixed
        // point, further iterations reproduce exactly these values
        if (__all(qb[0] == was0 && qb[1] == was1)) break;
    }

    if (__all(err >= bound)) return;

    int32_t aq[2];
    uint32_t aqb[2];
    const uint32_t vp[4] = {ln.tx.pl[0][0], ln.tx.pl[0][1], ln.tx.pl[0][2], ln.tx.pl[0][3]};
    const uint32_t vg[4] = {ln.tx.pl[1][0], ln.tx.pl[1][1], ln.tx.pl[1][2], ln.tx.pl[1][3]};
    const uint32_t vb[4] = {ln.tx.pl[2][0], ln.tx.pl[2][1], ln.tx.pl[2][2], ln.tx.pl[2][3]};
    const uint32_t va[4] = {ln.tx.pl[3][0], ln.tx.pl[3][1], ln.tx.pl[3][2], ln.tx.pl[3][3]};
    uint32_t vsel[4];
    #pragma unroll
    for (int i = 0; i < 4; i++) vsel[i] = (rotation == 0) ? vp[i] : ((rotation == 1) ? vg[i] : ((rotation == 2) ? vb[i] : va[i]));
    err += encode_scalar<ABITS, AEPB>(aqb, aq, ln.tx, rotation, vsel, S.refineIterations_channel, ln.T, ln.pal);

    if (err < best_err) {
        #pragma unroll
        for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) best.q[i][p] = q[i][p];
        best.qb[0] = qb[0]; best.qb[1] = qb[1];
        best.aq[0] = aq[0]; best.aq[1] = aq[1];
        best.aqb[0] = aqb[0]; best.aqb[1] = aqb[1];
        best.rotation = rotation;
        best.swap = SWAP;
        best_err = err;
    }
}

__device__ __forceinline__ void clear(Dual& d)
{
    #pragma unroll
    for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) d.q[i][p] = 0;
    d.qb[0] = d.qb[1] = d.aqb[0] = d.aqb[1] = 0u;
    d.aq[0] = d.aq[1] = 0; d.rotation = 0; d.swap = 0;
}

// The reference tries all mode 4 candidates (rotation-major, swap-minor), commits, then all mode 5 candidates.
// Here one pass over the rotations serves both: a rotation's line fit does not depend on the mode.  Mode 5's
// winner is found independently (first strict minimum in rotation order) and admitted afterwards against the
// error left by mode 4 -- the same block the sequential scan produces.
// Rotations r0 .. r1-1 (the wide path gives each rotation its own wave).
// `which`: bit 0 = mode 4 without index swap, bit 1 = mode 4 swapped, bit 2 = mode 5 (the wide path runs them as separate tasks).
__device__ __forceinline__ void modes_45_scan(Lane& ln, const bc7_enc_settings& S, int r0, int r1, Dual& best4, int32_t& err4, Dual& best5, int32_t& err5,
                                              int which = 7)
{
    for (int r = r0; r < r1; r++) {
        // rotated colour block: channel r is replaced by alpha (RGBA profile) or 255 (RGB profile);
        // the displaced channel is coded separately
        Tex rot;
        const uint32_t sh8 = 8u * (uint32_t)r;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const uint32_t w = ln.tx.w[k];
            uint32_t v = w;
            if (r < 3) {
                const uint32_t fill = (S.channels == 4) ? (w >> 24) : 255u;
                v = (w & ~(255u << sh8)) | (fill << sh8);
            }
            rot.w[k] = v;
        }
#pragma unroll
        for (int c = 0; c < 4; c++)
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const uint32_t fill = (S.channels == 4) ? ln.tx.pl[3][d] : 0xffffffffu;
                rot.pl[c][d] = (c == r && r < 3) ? fill : ln.tx.pl[c][d];
            }
        IStats<3> st;
        stats_int<3>(st, rot.pl, whole_block());
        float fit[2][4];
        fit[0][3] = 0.f; fit[1][3] = 0.f;
        fit_line<3>(fit, rot, 0xffffu, st, rcp_of_count(16), ln.T);
        const int32_t tt = st.m[0] + st.m[4] + st.m[7];          // |texel|^2 summed over the rotated colour block
        if (which & 1) try_dual<4, 0>(best4, err4, err4, ln, rot, fit, tt, S, r);
        if (which & 2) try_dual<4, 1>(best4, err4, err4, ln, rot, fit, tt, S, r);
        // mode 5's winner is admitted only below the error mode 4 leaves (modes_45; the wide path's commit order), and err4 only
        // falls from here on: a mode 5 candidate at or above it can neither be admitted nor hide one that could
        if (which & 4) try_dual<5, 0>(best5, err5, min(err5, err4), ln, rot, fit, tt, S, r);
    }
}

__device__ __forceinline__ void modes_45(Lane& ln, const bc7_enc_settings& S)      // [kernel.ispc:1623-1655]
{
    Dual best4, best5;
    clear(best4); clear(best5);
    int32_t err4 = ln.best_err, err5 = ERR_MAX;
    modes_45_scan(ln, S, S.mode45_channel0, S.channels, best4, err4, best5, err5);
    if (err4 < ln.best_err) { ln.best_err = err4; ln.improved = true; emit_dual<4>(ln.best, best4); }
    if (err5 < ln.best_err) { ln.best_err = err5; ln.improved = true; emit_dual<5>(ln.best, best5); }
}

// ---- mode 6: one subset, RGBA, 4-bit indices                                     [kernel.ispc:1657-1689]
template <int CH>
__device__ __forceinline__ void mode_6(Lane& ln, const bc7_enc_settings& S)
{
    const SubsetMask all = whole_block();
    IStats<CH> st;
    stats_int<CH>(st, ln.tx.pl, all);
    float ep[2][4];
    int32_t q[2][4], d[2][4];
    uint32_t qb[2];
    Segment sg[3];
    ep[0][3] = 0.f; ep[1][3] = 0.f;
    fit_line<CH>(ep, ln.tx, 0xffffu, st, rcp_of_count(16), ln.T);
    if (CH == 3) { ep[0][3] = 255.f; ep[1][3] = 255.f; }
    quant_mode<6, true>(q, d, ep, CH);
#if ITW_BC7_LANE_PAL
    // the sixteen decoded colours once into the lane's LDS column, then one projection, one 16-byte read and two dot products
    // per texel instead of decoding both neighbouring levels with packed 16-bit math per texel (bc7_exact.hpp)
    const int32_t tt = block_norm2<CH>(ln.tx.pl);
    PalSegment ps = build_palette<4, CH, TPB>(ln.pal, d);
    int32_t err = select_block_pal<4, CH, TPB>(qb, ln.tx, ps, ln.pal, tt);
    (void)sg;
#else
    sg[0] = make_segment<4, CH>(d); sg[1] = sg[0]; sg[2] = sg[0];
    int32_t err = select_block<4, CH, 1>(qb, ln.tx, sg, 0u);
#endif
    const int iters = S.refineIterations[6];
    for (int it = 0; it < iters; it++) {
        const uint32_t was0 = qb[0], was1 = qb[1];
        refit_line<4, CH>(ep, ln.tx.pl, qb, all, ln.T);
        quant_mode<6, false>(q, d, ep, CH);
#if ITW_BC7_LANE_PAL
        ps = build_palette<4, CH, TPB>(ln.pal, d);
        err = select_block_pal<4, CH, TPB>(qb, ln.tx, ps, ln.pal, tt);
#else
        sg[0] = make_segment<4, CH>(d);
        err = select_block<4, CH, 1>(qb, ln.tx, sg, 0u);
#endif
        if (__all(qb[0] == was0 && qb[1] == was1)) break;         // fixed point of the iteration (kernel.ispc:1672-1677)
    }
    if (err < ln.best_err) {
        ln.best_err = err;
        ln.improved = true;
        emit_mode6(ln.best, q, qb);
    }
}

// ---- kernels ----------------------------------------------------------------------------------------
// Per multi-subset family two kernels: SEARCH (scan of the shapes; low register pressure, runs at 3-4 waves per SIMD)
// leaves each mode's winner {indices, error, shape} in the workspace, FINISH refines the winners and lets them compete
// for the block.  Modes 4/5/6 are one kernel.  Splitting keeps the register allocation of the scan independent of the
// (per-lane shape, fully unrolled) refinement code.
template <bool VEC16>
__device__ __forceinline__ void load_block(Tex& tx, const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t b)
{
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    const uint8_t* p = src + (int64_t)yy * 4 * stride + (int64_t)xx * 16;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        if (VEC16) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + y * stride);
            tx.w[y * 4 + 0] = v.x; tx.w[y * 4 + 1] = v.y; tx.w[y * 4 + 2] = v.z; tx.w[y * 4 + 3] = v.w;
        } else {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p + y * stride);
            #pragma unroll
            for (int x = 0; x < 4; x++) tx.w[y * 4 + x] = q[x];
        }
    }
    tx.make_planar();
}

// Round 5: the blocks a finish phase lists for a later scan also leave their 64 B of texels in a compact buffer, in list order, so the
// list scans (up to 8 shares per block) and their refinement read contiguous lines instead of gathering 64 B out of every 128 B line
__device__ __forceinline__ void load_block_compact(Tex& tx, const uint4* __restrict__ compact, int32_t slot)
{
    const uint4* p = compact + (int64_t)slot * 4;
#pragma unroll
    for (int y = 0; y < 4; y++) { const uint4 v = p[y]; tx.w[y * 4 + 0] = v.x; tx.w[y * 4 + 1] = v.y; tx.w[y * 4 + 2] = v.z; tx.w[y * 4 + 3] = v.w; }
    tx.make_planar();
}

// winners of a family's two modes: [slot][block] x {err, shape}  (8 B; the workspace keeps its 16 B slots)
__device__ __forceinline__ void store_win(uint4* __restrict__ wins, int32_t nblocks, int slot, int32_t b, const Win& w)
{
    reinterpret_cast<uint2*>(wins)[(int64_t)slot * nblocks + b] = make_uint2((uint32_t)w.err, (uint32_t)w.shape);
}
__device__ __forceinline__ void load_win(Win& w, const uint4* __restrict__ wins, int32_t nblocks, int slot, int32_t b)
{
    const uint2 v = reinterpret_cast<const uint2*>(wins)[(int64_t)slot * nblocks + b];
    w.err = (int32_t)v.x; w.shape = (int32_t)v.y; w.key = -1;
}

// Register budgets (waves per SIMD) were picked by measurement on MI355X: the scans are bound by dependent-issue
// and LDS latency at 2 waves, and tolerate a few spilled cold values to reach 3-4.  Re-swept in round 6 under the max-ILP schedule
// (profiles/r06_rejected_experiments.txt, item 10): only the unranked mode 7 scan moved, 3 -> 4 (alpha_slow -3...-5 %).
#ifndef SW02
#define SW02 4
#endif
#ifndef SW13
#define SW13 4
#endif
#ifndef SWALL
#define SWALL 4
#endif
#ifndef SW7
#define SW7 4
#endif
#ifndef SWR
#define SWR 2
#endif
#ifndef SWR2
#define SWR2 3
#endif
#ifndef SWR27
#define SWR27 2
#endif
#ifndef FW
#define FW 3
#endif
#ifndef FW456
#define FW456 2
#endif
__host__ __device__ constexpr int search_waves(int family, int ranked) { return ranked == 2 ? (family == F_MODE7 ? SWR27 : SWR2) : ranked ? SWR : (family == F_MODE7 ? SW7 : (family == F_MODES02 ? SW02 : SW13)); }
__host__ __device__ constexpr int finish_waves(int family) { return family == F_MODES13 ? FW : FW456; }

template <int FAMILY, int RANKED, bool VEC16>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(search_waves(FAMILY, RANKED), search_waves(FAMILY, RANKED))))
bc7_search_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks,
                  uint4* __restrict__ wins, const bc7_enc_settings S)
{
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    extern __shared__ int32_t s_keys[];            // 64 * TPB keys, only allocated for RANKED == 1
    // per-lane palettes: table-order scans 8 + 4 levels (24 KiB); ranked lists <= 16: 2 subsets x 8 levels (32 KiB); longer lists: none
    __shared__ uint2 s_pal[RANKED == 0 ? 12 * TPB : (RANKED == 2 && ITW_BC7_LANE_PAL) ? 16 * TPB : 1];
    Lane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB);
    __syncthreads();
    const int32_t gid = blockIdx.x * TPB + threadIdx.x;
    const bool live = gid < nblocks;
    const int32_t b = live ? gid : nblocks - 1;    // idle lanes of the last workgroup redo its last block, store nothing
    ln.keys = s_keys + threadIdx.x;
    ln.pal = s_pal + ((RANKED == 0 || (RANKED == 2 && ITW_BC7_LANE_PAL)) ? threadIdx.x : 0);
    load_block<VEC16>(ln.tx, src, stride, blocks_x, b);

    Win wa, wb;
    if (FAMILY == F_MODES02) search_02(ln, S, wa, wb);
    if (FAMILY == F_MODES13) search_two_subset<false, 3, RANKED>(ln, S, wa, wb);
    if (FAMILY == F_MODE7) {
        if (S.channels == 4) search_two_subset<true, 4, RANKED>(ln, S, wa, wb); else search_two_subset<true, 3, RANKED>(ln, S, wa, wb);
    }
    if (live) {
        store_win(wins, nblocks, 0, b, wa);
        if (FAMILY != F_MODE7) store_win(wins, nblocks, 1, b, wb);
    }
}

template <int FAMILY, bool VEC16>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(finish_waves(FAMILY), finish_waves(FAMILY))))
bc7_finish_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks,
                  uint8_t* __restrict__ dst, int32_t* __restrict__ err_ws, const uint4* __restrict__ wins,
                  const bc7_enc_settings S, const int first)
{
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    // mode 4/5 vector part: one palette per lane (16 KiB); refinement of a family's winners: one per subset (pairs x levels)
    __shared__ uint2 s_pal[FAMILY == F_MODES456 ? (ITW_BC7_LANE_PAL ? 16 : 8) * TPB : (ITW_BC7_LANE_PAL ? (FAMILY == F_MODES02 ? 24 : 16) * TPB : 1)];
    Lane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB);
    __syncthreads();
    const int32_t gid = blockIdx.x * TPB + threadIdx.x;
    const bool live = gid < nblocks;
    const int32_t b = live ? gid : nblocks - 1;
    ln.keys = nullptr;
    ln.pal = s_pal + ((FAMILY == F_MODES456 || ITW_BC7_LANE_PAL) ? threadIdx.x : 0);
    load_block<VEC16>(ln.tx, src, stride, blocks_x, b);

    ln.best_err = first ? ERR_MAX : err_ws[b];
    ln.best[0] = ln.best[1] = ln.best[2] = ln.best[3] = 0u;
    ln.improved = (first != 0);                    // the first family always defines the block
    ln.opaque_err = 0;                                                             // kernel.ispc:1267-1277
    if (S.channels == 4) {
        uint32_t e = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) { const uint32_t x = ~ln.tx.pl[3][d]; e = udot4(x, x, e); }   // (255 - a)^2, bytewise
        ln.opaque_err = (int32_t)e;
    }

    if (FAMILY == F_MODES02) {
        Win w;
        load_win(w, wins, nblocks, 0, b);
        refine_and_commit<0>(ln, w, S.refineIterations[0], S.channels);
        if (!S.skip_mode2) { load_win(w, wins, nblocks, 1, b); refine_and_commit<2>(ln, w, S.refineIterations[2], S.channels); }
    }
    if (FAMILY == F_MODES13) {
        Win w;
        if (S.fastSkipTreshold_mode1 > 0) { load_win(w, wins, nblocks, 0, b); refine_and_commit<1>(ln, w, S.refineIterations[1], S.channels); }
        if (S.fastSkipTreshold_mode3 > 0) { load_win(w, wins, nblocks, 1, b); refine_and_commit<3>(ln, w, S.refineIterations[3], S.channels); }
    }
    if (FAMILY == F_MODE7) {
        Win w;
        load_win(w, wins, nblocks, 0, b);
        refine_and_commit<7>(ln, w, S.refineIterations[7], S.channels);
    }
    if (FAMILY == F_MODES456) {
        if (S.mode_selection[2]) modes_45(ln, S);
        if (S.mode_selection[3]) { if (S.channels == 4) mode_6<4>(ln, S); else mode_6<3>(ln, S); }
    }

    if (live && ln.improved) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (int64_t)b * 16);
        if (VEC16) *reinterpret_cast<uint4*>(d) = make_uint4(ln.best[0], ln.best[1], ln.best[2], ln.best[3]);
        else { d[0] = ln.best[0]; d[1] = ln.best[1]; d[2] = ln.best[2]; d[3] = ln.best[3]; }
        err_ws[b] = ln.best_err;
    }
}

enum WideKind { WK_SCAN02 = 0, WK_SCAN13 = 1, WK_SCAN7 = 2 };          // scan tasks of the fused and the wide path
__host__ __device__ constexpr int wide_win_slot(int mode) { return mode == 0 ? 0 : mode == 2 ? 1 : mode == 1 ? 2 : mode == 3 ? 3 : 4; }   // winner rows: m0 m2 m1 m3 m7

// the ranking's view of the whole block, for rank keys evaluated at merge time (same construction as search_two_subset)
template <int FIT_CH, int RANK_CH>
__device__ __forceinline__ void whole_block_rank_stats(Stats<RANK_CH>& rfull, const Tex& tx)
{
    IStats<FIT_CH> full;
    stats_int<FIT_CH>(full, tx.pl, whole_block());
    IStats<RANK_CH> t;
    #pragma unroll
    for (int i = 0; i < 10; i++) t.m[i] = full.m[i];
    #pragma unroll
    for (int i = 0; i < 4; i++) t.s[i] = full.s[i];
    t.n = 16;
    stats_float<RANK_CH>(rfull, t);
}

template <int FIT_CH, int RANK_CH>
__device__ __forceinline__ int32_t merge_key(const Win& w, const Tex& tx, const SeedTables& T)
{
    if (w.key >= 0) return w.key;
    Stats<RANK_CH> rfull;
    whole_block_rank_stats<FIT_CH, RANK_CH>(rfull, tx);
    return rank_key<RANK_CH>(w.shape, tx, rfull, T);
}

// ---- FUSED deep path: whole surfaces in two launches --------------------------------------------------------------------
// The kernels above run as up to seven launches, each re-reading the 64 B block from HBM and handing winners, best error
// and the current block to the next through the workspace: 5.4x the algorithmic bytes for a `slow` call.  Nothing in
// that chain needs a kernel boundary except the register budgets (scans: 4 waves per SIMD; refinement: 2).  So:
//   bc7_scan_all    ONE launch for the scans of the three-channel families {0,2} and {1,3} (mode 7, whose four-channel
//                   fits need more registers, gets its own).  Workgroup ids are mapped XCD-aware: MI355X deals consecutive
//                   workgroups round-robin to its 8 XCDs, each with its own L2, so (chunk, family) = f(workgroup id) runs
//                   256 chunks of one family, then the SAME chunks of the other, every chunk on the same XCD both times
//                   -- the second family's texel loads hit that L2 instead of HBM, while a CU still runs one family's
//                   code for long stretches.  Winners leave as 4 bytes per mode: error (< 2^23: 16 texels x 4 channels
//                   x 255^2) << 7 | shape;
//   bc7_finish_all  ONE launch: each lane refines its modes' winners in the reference's order 0,2,1,3,7, then modes
//                   4,5,6, carrying best error and block in registers, and writes the block once.
// RGBA profiles (channels == 4) with both mode groups enabled run the same kernels in the order  scan 7 -> finish<1> (modes
// 7,4,5,6; leaves block + error) -> scan {0,2},{1,3} -> finish<2> (modes 0,2,1,3; replaces the block iff its error <= the
// alpha group's, which is the reference's first-strict-minimum over 0,2,1,3,7,4,5,6).  A three-channel mode's error includes
// sum (255 - alpha)^2, so waves whose 64 blocks all have that term above their alpha-group error skip the RGB scans and
// their refinement: on translucent content most of the call (alpha_slow 10.7 -> 4.7 ms at 4096^2), on opaque content nothing
// is skipped and the result is the same bytes either way.
// HBM bytes per block: 64 (scans, shared through L2) + 64 (finish) + 2 x 4 x modes + 16 = ~176: 2.1x algorithmic.
__device__ __forceinline__ uint32_t pack_win(const Win& w) { return w.err == ERR_MAX ? 0xffffffffu : (((uint32_t)w.err << 7) | ((uint32_t)w.shape & 127u)); }
__device__ __forceinline__ void unpack_win(Win& w, uint32_t v)
{
    if (v == 0xffffffffu) { w.err = ERR_MAX; w.shape = 0; } else { w.err = (int32_t)(v >> 7); w.shape = (int32_t)(v & 127u); }
    w.key = -1;
}

#ifndef LIST_SCAN_MAX_PARTS
#define LIST_SCAN_MAX_PARTS 8
#endif
// split scans of the fused path (bounded order, modes 1/3 over a short list): how many strided shares of the 64 shapes each listed block's
// scan is cut into so that a short list still fills the chip -- a function of the list length alone, computed alike by the scan and by
// the refinement that merges the shares.  parts x listed blocks <= nblocks: the shares' winners fit the mode's winner row, [part][slot].
__device__ __forceinline__ int32_t list_scan_parts(int32_t count, int32_t nblocks)
{
    const int32_t chunks = ((count + TPB - 1) / TPB + 7) & ~7;        // in eights: a chunk's shares stay on one XCD (below)
    if (chunks <= 0) return 1;
    const int32_t p = nblocks / (chunks * TPB);
    return p < 1 ? 1 : (p > LIST_SCAN_MAX_PARTS ? LIST_SCAN_MAX_PARTS : p);
}
// ordered argmin over the shares' winners of mode 1 / 3 (packed: the rank key is evaluated on a tie only, as in the wide path)
template <int FIT_CH, int RANK_CH>
__device__ __forceinline__ void merge_packed_parts(Win& w, Lane& ln, const uint32_t* __restrict__ row, int32_t count, int parts, int32_t slot)
{
    unpack_win(w, row[slot]);
    for (int p = 1; p < parts; p++) {
        Win x;
        unpack_win(x, row[(int64_t)p * count + slot]);
        if (x.err < w.err) { w = x; continue; }
        if (x.err != w.err || x.err == ERR_MAX) continue;
        w.key = merge_key<FIT_CH, RANK_CH>(w, ln.tx, ln.T); x.key = merge_key<FIT_CH, RANK_CH>(x, ln.tx, ln.T);
        if (x.key < w.key) w = x;
    }
}

struct ScanTasks { int n; int kind[3]; };      // WK_SCAN02 / WK_SCAN13 / WK_SCAN7 in launch order

// Round 5, the BANDS and the PILOT of the bounded order (launch_bc7): which chunks (runs of TPB blocks) of the surface a launch of
// bc7_scan_all / bc7_finish_all walks, and whether it runs at all.
//   kind 0  every chunk: workgroup i -> chunk i
//   kind 2  every eighth chunk (8 i + 3): what bc7_pilot_estimate looks at, for a probe that scans nothing else
//   kind 1  a BAND: the surface is cut into stripes of `stripe` chunks, even stripes are band 0, odd stripes band 1 (two interleaved
//           halves, so that both see all of the surface's content); workgroup i takes the i-th chunk of band `band`
// `gate` (optional): a device word written by bc7_pilot_estimate before this launch starts; the launch returns at once unless it holds
// `want` -- both mode orders are enqueued behind the pilot and the device picks one, without a host round trip.
struct ChunkSel { int32_t kind, stripe, band; const int32_t* gate; int32_t want; };
__device__ __forceinline__ int32_t sel_chunk(const ChunkSel& s, int32_t i)
{
    if (s.kind == 0) return i;
    if (s.kind == 2) return i * 8 + 3;
    const int32_t q = i / s.stripe;
    return (2 * q + s.band) * s.stripe + (i - q * s.stripe);
}

// FAM7 = false: the three-channel families {0,2} and {1,3} (4 waves per SIMD; ranked lists 3); FAM7 = true: the mode 7 scan
// alone (four-channel fits need more registers; sharing a kernel with the others made those spill.  Unranked 4 waves since the
// max-ILP schedule, ranked lists 2).
__host__ __device__ constexpr int scan_all_waves(bool ranked, bool fam7) { return fam7 ? (ranked ? SWR27 : SW7) : (ranked ? SWR2 : SWALL); }
template <bool VEC16, bool RANKED_LISTS, bool FAM7>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(scan_all_waves(RANKED_LISTS, FAM7), scan_all_waves(RANKED_LISTS, FAM7))))
bc7_scan_all(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, uint32_t* __restrict__ wins4,
             const bc7_enc_settings S, const ScanTasks tasks, const int ranked13, const int ranked7, const int32_t nchunks, const int32_t grain,
             const int32_t* __restrict__ rgb_list, const int32_t* __restrict__ rgb_count, const int32_t split, const ChunkSel sel,
             const uint4* __restrict__ compact)
{
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    __shared__ uint2 s_pal[((RANKED_LISTS && ITW_BC7_LANE_PAL) ? 16 : 12) * TPB];   // ranked lists: 2 subsets x 8 levels (3 waves per SIMD: 3 x 44 KiB)
    if (sel.gate && *sel.gate != sel.want) return;                  // the pilot chose the other order (whole grid: no barrier is pending)
    if (!RANKED_LISTS && split) {
        // bounded order: modes 1/3 (or mode 7: FAM7) over the list bc7_finish_all<.., 3 | 5 | 6> left, each listed block's 64 shapes cut
        // into `parts` strided shares (list_scan_parts) so that a short list still fills the chip; winners go to [part][slot] of the modes' rows
        const int32_t count = *rgb_count;
        const int32_t parts = list_scan_parts(count, nblocks);
        // workgroup w runs on XCD w % 8: the shares of a chunk are 8 workgroups apart, so they read its texels through the same L2
        const uint32_t wg = blockIdx.x;
        const int32_t chunk = (int32_t)((wg / (8u * (uint32_t)parts)) * 8u + (wg & 7u)), part = (int32_t)((wg >> 3) % (uint32_t)parts);
        if (chunk * TPB >= count) return;                                // whole workgroup: no barrier is pending
        Lane ln;
        ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB);
        __syncthreads();
        const int32_t gid = chunk * TPB + threadIdx.x;
        const bool live = gid < count;
        const int32_t slot = live ? gid : count - 1;
        ln.keys = nullptr;
        ln.pal = s_pal + threadIdx.x;
        if (compact) load_block_compact(ln.tx, compact, slot); else load_block<VEC16>(ln.tx, src, stride, blocks_x, rgb_list[slot]);
        Win wa, wb;
        if (FAM7) {
            search_two_subset<true, 4, 0>(ln, S, wa, wb, part, parts);             // RGBA profiles only (launch_bc7)
            if (live) wins4[(int64_t)wide_win_slot(7) * nblocks + (int64_t)part * count + slot] = pack_win(wa);
        } else {
            search_two_subset<false, 3, 0>(ln, S, wa, wb, part, parts);
            if (live) {
                wins4[(int64_t)wide_win_slot(1) * nblocks + (int64_t)part * count + slot] = pack_win(wa);
                wins4[(int64_t)wide_win_slot(3) * nblocks + (int64_t)part * count + slot] = pack_win(wb);
            }
        }
        return;
    }
    // `grain` chunks of one family, then the same chunks of the next family (grain is a multiple of 8, so a chunk's
    // families run on the same XCD: workgroup w -> XCD w % 8)
    const uint32_t w = blockIdx.x, per = (uint32_t)grain * (uint32_t)tasks.n, group = w / per, r = w % per;
    const int t = (int)(r / (uint32_t)grain);
    const int32_t chunk_i = (int32_t)(group * (uint32_t)grain + r % (uint32_t)grain);
    if (chunk_i >= nchunks) return;                                  // whole workgroup: no barrier is pending
    const int32_t chunk = sel_chunk(sel, chunk_i);                   // nchunks counts the chunks `sel` selects
    // RGBA profile, alpha-capable modes already encoded (bc7_finish_all<.., 1>): an RGB-only mode's error includes
    // sum (alpha - 255)^2 (kernel.ispc:1267-1277, 1356), so where that term alone exceeds the alpha modes' best error no
    // three-channel mode can win or tie.  Round 3: finish<1> COMPACTS the blocks where an RGB mode can still win or tie into
    // rgb_list (any order: blocks are independent), and this kernel walks the list instead of the surface -- lanes are
    // filled with blocks that need the scans whatever their position, so a surface whose opaque and translucent blocks are
    // mixed at random pays for the opaque half only (round 2 skipped whole waves: nothing on such content).
    const int32_t nact = (!FAM7 && rgb_list) ? *rgb_count : nblocks;
    if (chunk * TPB >= nact) return;
    Lane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB);
    __syncthreads();
    const int32_t gid = chunk * TPB + threadIdx.x;
    const bool live = gid < nact;
    const int32_t slot = live ? gid : nact - 1;                      // idle lanes of the last workgroup redo its last block, store nothing
    const int32_t b = (!FAM7 && rgb_list) ? rgb_list[slot] : slot;
    ln.keys = nullptr;
    ln.pal = s_pal + threadIdx.x;
    load_block<VEC16>(ln.tx, src, stride, blocks_x, b);
    const int kind = tasks.kind[t];                                  // wave-uniform
    Win wa, wb;
    if (!FAM7 && kind == WK_SCAN02) {
        search_02(ln, S, wa, wb);
        if (live) { wins4[(int64_t)wide_win_slot(0) * nblocks + b] = pack_win(wa); if (!S.skip_mode2) wins4[(int64_t)wide_win_slot(2) * nblocks + b] = pack_win(wb); }
    } else if (!FAM7) {
        if (RANKED_LISTS && ranked13) search_two_subset<false, 3, 2>(ln, S, wa, wb); else search_two_subset<false, 3, 0>(ln, S, wa, wb);
        if (live) { wins4[(int64_t)wide_win_slot(1) * nblocks + b] = pack_win(wa); wins4[(int64_t)wide_win_slot(3) * nblocks + b] = pack_win(wb); }
    } else {
        if (S.channels == 4) { if (RANKED_LISTS && ranked7) search_two_subset<true, 4, 2>(ln, S, wa, wb); else search_two_subset<true, 4, 0>(ln, S, wa, wb); }
        else                 { if (RANKED_LISTS && ranked7) search_two_subset<true, 3, 2>(ln, S, wa, wb); else search_two_subset<true, 3, 0>(ln, S, wa, wb); }
        if (live) wins4[(int64_t)wide_win_slot(7) * nblocks + b] = pack_win(wa);
    }
}

// appends block `b` of the lanes with `need` to a list: one atomic per wave, ranks within the wave by ballot (any order: blocks are independent)
// returns the block's position in the list (-1: not appended)
__device__ __forceinline__ int32_t append_to_list(int32_t* __restrict__ list, int32_t* __restrict__ count, bool need, int32_t b)
{
    const unsigned long long m = __ballot(need);
    int32_t pos = -1;
    if (m) {
        const int lane = (int)(threadIdx.x & 63u);
        int32_t base = 0;
        if (lane == 0) base = atomicAdd(count, (int32_t)__popcll(m));
        base = __shfl(base, 0);
        if (need) { pos = base + (int32_t)__popcll(m & ((1ull << lane) - 1ull)); list[pos] = b; }
    }
    return pos;
}

// PHASE 0: every mode, reference order (RGB profiles).  RGBA profiles run in two phases so that the three-channel modes can be
// skipped where they cannot win: PHASE 1 = the alpha-capable modes 7,4,5,6 (their relative order kept), leaves the block and
// its error; PHASE 2 = modes 0,2,1,3 for the waves that still need them, then the reference's choice between the two groups:
// the first strict minimum over 0,2,1,3,7,4,5,6 is the RGB group's winner iff its error <= the alpha group's.
//
// Round 4, the BOUNDED order (profiles that scan every two-subset shape: `slow`, `alpha_slow`): modes 1 and 3 run LAST, and only for the
// blocks where some two-subset shape's exact lower bound (two_subset_bound, bc7_exact.hpp) is still below what the other modes achieved.
//   PHASE 3 (RGB profiles)   modes 0,2 then 4,5,6 -> block, and the incumbent `inc` modes 1/3 have to get strictly below: the error,
//                            + 1 when it belongs to mode 4/5/6 (those come after 1/3 in the reference's order, so a tie goes to 1/3);
//                            then the list of blocks with min over shapes of the bound < inc;
//   PHASE 5 (RGBA profiles)  the same on the list PHASE 1 left: modes 0,2, the choice against the alpha group (ties to 0/2), inc, list;
//   PHASE 4                  modes 1,3 over that list, starting from best_err = inc: the block is replaced iff one of them gets below.
// RGBA profiles whose mode 7 scans every shape too (`alpha_slow`) bound it with the same number (a mode 7 encoding's colour part is one more
// set of rounded points of a segment per subset; its error has no opaque term):
//   PHASE 6                  modes 4,5,6 alone -> block, error; the list of blocks where a three-channel mode can still win or tie, as
//                            PHASE 1 leaves it (judged against modes 4,5,6 only: a longer list, the same final comparison), and for the
//                            other blocks inc = error + 1 and a place on mode 7's list (translucent content: not bounded, mode 7 is its mode);
//   PHASE 5                  as above, and it also appends the blocks whose mode 7 bound is below inc to mode 7's list;
//   PHASE 4                  as above, and leaves inc = its error where it replaced the block (mode 7 comes after 1/3: strictly below);
//   PHASE 7                  mode 7 over its list from best_err = inc.
// The final block is the reference's first strict minimum over 0,2,1,3,7,4,5,6 either way: a mode's result enters only through
// `err < best_err`, the groups keep their internal order, and a skipped block's modes 1/3 cannot reach inc whatever shape their scan would
// have picked (every encoding of every shape is bounded).
// Round 5: mode 6 under an RGB profile (kernel.ispc:1657-1689, channels == 3) decodes every texel to a ROUNDED point of ONE segment in RGB,
// so its error is >= (sqrt(R) - sqrt(3)/2 sqrt(16))_+^2 with R the block's residual about its best line (subset_residual_bound: the same
// number the two-subset bound is built from, for the whole block), and it replaces the block only on a strict `<`.  Where that bound has
// already been reached by the modes before it for all 64 blocks of the wave, mode 6 is not run (per-lane skipping saves nothing on a SIMT
// machine).  profiles/history/r05/r05_bc7_bound456_study.txt: 100 % of the bench surface's blocks (noise in every block: three subsets beat one line),
// 56-64 % of a photograph's; modes 4/5 cannot be bounded out this way (15 % / 1 %: their scalar channel absorbs the noise).
__device__ __forceinline__ bool mode6_cannot_win(Lane& ln)
{
    if (__all(ln.best_err == 0)) return true;
    IStats<3> full;
    stats_int<3>(full, ln.tx.pl, whole_block());
    const float d = __builtin_amdgcn_sqrtf(subset_residual_bound(full) * rcp_of_count(16)) * 0.999999f - BOUND_SLACK3[16];
    const float e = fmaxf(d, 0.0f);
    const float lb = e * e * 0.999999f;
    return __all(lb >= (float)ln.best_err);                        // errors are integers below 2^24: exact as floats
}

#ifndef FINISH_ALL_WAVES
#define FINISH_ALL_WAVES 2
#endif
#ifndef BOUND_BAIL_AFTER
#define BOUND_BAIL_AFTER 4
#endif
#ifndef BOUND_BAIL_LANES
#define BOUND_BAIL_LANES 40
#endif
template <bool VEC16, int PHASE>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(FINISH_ALL_WAVES, FINISH_ALL_WAVES)))
bc7_finish_all(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, uint8_t* __restrict__ dst,
               const uint32_t* __restrict__ wins4, const bc7_enc_settings S, int32_t* __restrict__ inc_err,
               const int32_t* __restrict__ in_list, const int32_t* __restrict__ in_count, int32_t* __restrict__ out_list, int32_t* __restrict__ out_count,
               int32_t* __restrict__ out_list7, int32_t* __restrict__ out_count7, const ChunkSel sel,
               const uint4* __restrict__ in_compact, uint4* __restrict__ out_compact)
{
    constexpr bool LISTED = PHASE == 2 || PHASE == 4 || PHASE == 5 || PHASE == 7;   // walks a compacted list of an earlier phase
    constexpr bool DO02 = PHASE == 0 || PHASE == 2 || PHASE == 3 || PHASE == 5;
    constexpr bool DO13 = PHASE == 0 || PHASE == 2 || PHASE == 4;
    constexpr bool DO7 = PHASE == 0 || PHASE == 1;
    constexpr bool DO456 = PHASE == 0 || PHASE == 1 || PHASE == 3 || PHASE == 6;
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    __shared__ uint2 s_pal[(ITW_BC7_LANE_PAL ? LANE_PAL_LEVELS : 8) * TPB];   // refinement: a palette per subset of the lane's winner
    if (sel.gate && *sel.gate != sel.want) return;                   // the pilot chose the other order (whole grid: no barrier is pending)
    const int32_t nact = LISTED ? *in_count : nblocks;
    const int32_t chunk = LISTED ? (int32_t)blockIdx.x : sel_chunk(sel, (int32_t)blockIdx.x);
    if (chunk * TPB >= nact) return;                                 // whole workgroup, before any barrier (a band's grid is whole stripes)
    Lane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB);
    __syncthreads();
    const int32_t gid = chunk * TPB + threadIdx.x;
    const bool live = gid < nact;
    const int32_t slot = live ? gid : nact - 1;
    const int32_t b = LISTED ? in_list[slot] : slot;
    ln.keys = nullptr;
    ln.pal = s_pal + threadIdx.x;
    if (LISTED && in_compact) load_block_compact(ln.tx, in_compact, slot); else load_block<VEC16>(ln.tx, src, stride, blocks_x, b);
    ln.best_err = ERR_MAX;
    ln.best[0] = ln.best[1] = ln.best[2] = ln.best[3] = 0u;
    ln.improved = false;
    ln.opaque_err = 0;                                                             // kernel.ispc:1267-1277
    if (S.channels == 4) {
        uint32_t e = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) { const uint32_t x = ~ln.tx.pl[3][d]; e = udot4(x, x, e); }
        ln.opaque_err = (int32_t)e;
    }
    const bool on13 = S.mode_selection[1];
    int32_t e_alpha = ERR_MAX;
    if (PHASE == 2 || PHASE == 5) e_alpha = inc_err[b];      // every listed block has opaque_err <= e_alpha: its winners exist
    if (PHASE == 4 || PHASE == 7) ln.best_err = inc_err[b];  // what modes 1/3 (mode 7) have to get strictly below
    Win w;
    if (DO02 && S.mode_selection[0]) {
        unpack_win(w, wins4[(int64_t)wide_win_slot(0) * nblocks + b]);
        refine_and_commit<0>(ln, w, S.refineIterations[0], S.channels);
        if (!S.skip_mode2) { ln.tx.fence(); unpack_win(w, wins4[(int64_t)wide_win_slot(2) * nblocks + b]); refine_and_commit<2>(ln, w, S.refineIterations[2], S.channels); }
    }
    const int32_t e02 = ln.best_err;
    if (DO13 && PHASE != 4) {
        if (on13 && S.fastSkipTreshold_mode1 > 0) { ln.tx.fence(); unpack_win(w, wins4[(int64_t)wide_win_slot(1) * nblocks + b]); refine_and_commit<1>(ln, w, S.refineIterations[1], S.channels); }
        if (on13 && S.fastSkipTreshold_mode3 > 0) { ln.tx.fence(); unpack_win(w, wins4[(int64_t)wide_win_slot(3) * nblocks + b]); refine_and_commit<3>(ln, w, S.refineIterations[3], S.channels); }
    }
    if (PHASE == 4) {                                        // the list scan left [part][slot] winners (bc7_scan_all, split)
        const int parts = list_scan_parts(nact, nblocks);
        if (S.fastSkipTreshold_mode1 > 0) { ln.tx.fence(); merge_packed_parts<3, 3>(w, ln, wins4 + (int64_t)wide_win_slot(1) * nblocks, nact, parts, slot); refine_and_commit<1>(ln, w, S.refineIterations[1], S.channels); }
        if (S.fastSkipTreshold_mode3 > 0) { ln.tx.fence(); merge_packed_parts<3, 3>(w, ln, wins4 + (int64_t)wide_win_slot(3) * nblocks, nact, parts, slot); refine_and_commit<3>(ln, w, S.refineIterations[3], S.channels); }
    }
    if (PHASE == 7) {                                        // the list scan left [part][slot] winners
        const int parts = list_scan_parts(nact, nblocks);
        ln.tx.fence();
        merge_packed_parts<4, 4>(w, ln, wins4 + (int64_t)wide_win_slot(7) * nblocks, nact, parts, slot);
        refine_and_commit<7>(ln, w, S.refineIterations[7], S.channels);
    }
    if (DO7 && on13 && S.fastSkipTreshold_mode7 > 0) { ln.tx.fence(); unpack_win(w, wins4[(int64_t)wide_win_slot(7) * nblocks + b]); refine_and_commit<7>(ln, w, S.refineIterations[7], S.channels); }
    if (DO456) {
        ln.tx.fence();
        if (S.mode_selection[2]) modes_45(ln, S);
        if (S.mode_selection[3]) {
            if (S.channels == 4) mode_6<4>(ln, S);
            else if (!mode6_cannot_win(ln)) mode_6<3>(ln, S);
        }
    }
    if (PHASE == 1) {
        if (live) inc_err[b] = ln.best_err;
        // the blocks where a three-channel mode can still win or tie (its error carries sum (255 - a)^2): appended to the list the
        // RGB scans and finish<2> walk -- one atomic per wave, ranks within the wave by ballot
        append_to_list(out_list, out_count, live && ln.opaque_err <= ln.best_err, b);
    }
    if (PHASE == 6) {
        const bool rgb = ln.opaque_err <= ln.best_err;               // a three-channel mode can still win or tie (PHASE 1's rule)
        if (live) inc_err[b] = rgb ? ln.best_err : ln.best_err + 1; // PHASE 5 reads the error; PHASE 7 reads what mode 7 has to get below
        append_to_list(out_list, out_count, live && rgb, b);
        append_to_list(out_list7, out_count7, live && !rgb, b);
    }
    if (PHASE == 3 || PHASE == 5) {
        // incumbent of modes 1/3 (see the header), then the blocks whose modes 1/3 can still get below it
        int32_t inc;
        if (PHASE == 3) inc = ln.best_err + (ln.best_err < e02 ? 1 : 0);
        else            inc = (e02 <= e_alpha) ? e02 : e_alpha + 1;
        if (live) inc_err[b] = inc;
        ln.tx.fence();
        IStats<3> full;
        stats_int<3>(full, ln.tx.pl, whole_block());
        const float lim = (float)(inc - ln.opaque_err) - 0.5f;      // errors are integers: a bound above inc - 1 already rules the shape out
        const float lim7 = (float)inc - 0.5f;                        // mode 7 carries no opaque term
        const bool with7 = PHASE == 5 && out_list7 != nullptr;       // wave-uniform
        bool need = !live, need7 = !live || !with7;                  // idle lanes never hold the loop up
#pragma unroll 1
        for (int shape = 0; shape < 64; shape++) {
            if (__all(need && need7)) break;
            // a wave most of whose blocks need the modes anyway stops bounding the others (visiting a block is always allowed):
            // content the bound does not separate pays for a few shapes, not for 64
            if (shape == BOUND_BAIL_AFTER && __popcll(__ballot(need && need7)) >= BOUND_BAIL_LANES) { need = true; need7 = true; break; }
            const float lb = two_subset_bound(shape, ln.tx.pl, full);
            need = need || !(lb >= lim);
            need7 = need7 || !(lb >= lim7);
        }
        const int32_t pos = append_to_list(out_list, out_count, live && need, b);
        if (out_compact && pos >= 0) {                              // the listed block's texels, in list order (load_block_compact)
            uint4* c = out_compact + (int64_t)pos * 4;
#pragma unroll
            for (int y = 0; y < 4; y++) c[y] = make_uint4(ln.tx.w[y * 4 + 0], ln.tx.w[y * 4 + 1], ln.tx.w[y * 4 + 2], ln.tx.w[y * 4 + 3]);
        }
        if (with7) append_to_list(out_list7, out_count7, live && need7, b);
    }
    bool store = live;
    if (PHASE == 2 || PHASE == 5) store = store && ln.best_err <= e_alpha;   // else the alpha group's block stands (ties go to the earlier, RGB, group)
    if (PHASE == 4 || PHASE == 7) store = store && ln.improved;
    if (PHASE == 4 && store) inc_err[b] = ln.best_err;         // mode 7 (PHASE 7, RGBA profiles) comes after modes 1/3: strictly below them
    if (store) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (int64_t)b * 16);
        if (VEC16) *reinterpret_cast<uint4*>(d) = make_uint4(ln.best[0], ln.best[1], ln.best[2], ln.best[3]);
        else { d[0] = ln.best[0]; d[1] = ln.best[1]; d[2] = ln.best[2]; d[3] = ln.best[3]; }
    }
}

// The PILOT of the bounded order (round 5).  Runs behind band 0's {0,2} scan on every eighth chunk of that band (1/16 of the surface, spread
// over it by the bands' stripes): with the scan's winners as the incumbent -- before refinement and before modes 4,5,6, so the true
// incumbent is lower and the true list shorter: an ESTIMATE from above (bench surface: 50 % estimated, 30 % listed; a photograph: 98 % and
// 94 %) -- it counts the blocks some two-subset shape's bound can still get under, with bc7_finish_all<3>'s rules, and the workgroup
// that finishes last turns the two counts into the device word the
// gated launches read: 1 = few enough blocks will need modes 1/3, the bounded order pays; 0 = nearly all will, the reference's order (one
// scan of the family over the band, one refinement kernel) is cheaper.  Nothing it computes reaches the output.
template <bool VEC16>
__global__ void __launch_bounds__(TPB)
bc7_pilot_estimate(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, const uint32_t* __restrict__ wins4,
                   const int skip_mode2, const ChunkSel sel, int32_t* __restrict__ counters /* listed, sampled, done */, const int32_t thr256,
                   int32_t* __restrict__ flag, int32_t* __restrict__ host_counts /* optional: pinned host memory, {listed, sampled} for the host */)
{
    const int32_t chunk = sel_chunk(sel, (int32_t)blockIdx.x * 8 + 3);
    if (chunk * TPB < nblocks) {                                     // workgroup-uniform
        const int32_t gid = chunk * TPB + threadIdx.x;
        const bool live = gid < nblocks;
        const int32_t b = live ? gid : nblocks - 1;
        Tex tx;
        load_block<VEC16>(tx, src, stride, blocks_x, b);
        Win w0, w2;
        unpack_win(w0, wins4[(int64_t)wide_win_slot(0) * nblocks + b]);
        unpack_win(w2, skip_mode2 ? 0xffffffffu : wins4[(int64_t)wide_win_slot(2) * nblocks + b]);
        const int32_t e = min(w0.err, w2.err);
        IStats<3> full;
        stats_int<3>(full, tx.pl, whole_block());
        const float lim = (float)e - 0.5f;
        bool need = !live;
#pragma unroll 1
        for (int shape = 0; shape < 64; shape++) {
            if (__all(need)) break;
            if (shape == BOUND_BAIL_AFTER && __popcll(__ballot(need)) >= BOUND_BAIL_LANES) { need = true; break; }
            need = need || !(two_subset_bound(shape, tx.pl, full) >= lim);
        }
        const unsigned long long ml = __ballot(live && need), mt = __ballot(live);
        if ((threadIdx.x & 63u) == 0u) { atomicAdd(counters + 0, (int32_t)__popcll(ml)); atomicAdd(counters + 1, (int32_t)__popcll(mt)); }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(counters + 2, 1) == (int32_t)gridDim.x - 1) {  // the last workgroup: every count has arrived
            __threadfence();
            const int64_t listed = atomicAdd(counters + 0, 0), sampled = atomicAdd(counters + 1, 0);
            *flag = (listed * 256 <= (int64_t)thr256 * sampled) ? 1 : 0;
            if (host_counts) { host_counts[0] = (int32_t)listed; host_counts[1] = (int32_t)sampled; __threadfence_system(); }
        }
    }
}

// ---- WIDE path: calls too small to fill the chip -----------------------------------------------------------------
// The kernels above give every block one lane and every mode family its own pair of launches, which is what fills
// 1024 SIMDs for a whole surface -- and what makes a 16 384-block call (the plugin's 0x40000-pixel slice,
// IntelPlugin.cpp:851) take a millisecond: 256 waves walk through five dependent launches while three quarters of the
// chip idle.  The reference's own answer to small work items is to cut them into bands for more threads
// (win32Threads.cpp:211-249); the GPU-side equivalent cuts the *search* instead:
//   phase 1  one launch whose blockIdx.y enumerates independent tasks -- the scan of each multi-subset family split
//            into `parts` strided shares of its candidate list (lanes stay blocks; another wave takes the other shapes).
//            A split scan leaves one winner per (mode, part).  Register budget of the scans: 4 waves per SIMD;
//   phase 2  one launch, blockIdx.y = task: per multi-subset mode an ORDERED ARGMIN over the parts' winners -- lowest
//            error, then the reference's strict-`<` tie rule (lowest table index for modes 0/2, lowest PCA rank key for
//            modes 1/3/7, kernel.ispc:1320, 1348, 1404-1409; keys are evaluated only on a tie) -- then the least-squares
//            refinement; next to them one task per mode 4/5 rotation and one for mode 6 (2 waves per SIMD);
//   phase 3  the modes' candidates {error, block} compete in the reference's order 0,2,1,3,7,4,5,6 with strict `<`
//            (kernel.ispc:1970-1977).
// Every candidate is evaluated by exactly the code of the deep path (same functions, same arithmetic), only by a different
// wave, so the emitted block is identical; tests run both paths on the same inputs (ITW_BC7_PATH=deep|wide).
constexpr int WIDE_SLOTS = 18;            // candidate slots in commit order: m0 m2 m1 m3 m7 m4[r0..r3][swap 0,1] m5[r0..r3] m6
constexpr int WIDE_SLOT_M4 = 5, WIDE_SLOT_M5 = 13, WIDE_SLOT_M6 = 17;
constexpr int WIDE_MAX_TASKS = 56;
constexpr int WIDE_MAX_PARTS = 16;
struct WideTasks { int n; uint8_t kind[WIDE_MAX_TASKS]; uint8_t part[WIDE_MAX_TASKS]; uint8_t parts[WIDE_MAX_TASKS]; };
struct WideModes { int n; uint8_t mode[20]; uint8_t parts[20]; };    // mode: 0,2,1,3,7 refine | 100 + 4 r + c: rotation r, candidate c (mode 4, mode 4 swapped, mode 5) | 6


// winners of split scans: [win_slot][part][block] x {err, shape, key, -}, `pstride` parts per slot
struct WideDims { int32_t nblocks; int32_t pstride; };
__device__ __forceinline__ void store_win_wide(uint4* __restrict__ wins, WideDims d, int slot, int part, int32_t b, const Win& w)
{
    wins[((int64_t)slot * d.pstride + part) * d.nblocks + b] = make_uint4((uint32_t)w.err, (uint32_t)w.shape, (uint32_t)w.key, 0u);
}
__device__ __forceinline__ void load_win_wide(Win& w, const uint4* __restrict__ wins, WideDims d, int slot, int part, int32_t b)
{
    const uint4 v = wins[((int64_t)slot * d.pstride + part) * d.nblocks + b];
    w.err = (int32_t)v.x; w.shape = (int32_t)v.y; w.key = (int32_t)v.z;
}
__device__ __forceinline__ void store_candidate(int32_t* __restrict__ cerr, uint4* __restrict__ cblk, int32_t nblocks, int slot, int32_t b,
                                                int32_t err, const uint32_t (&blk)[4])
{
    cerr[(int64_t)slot * nblocks + b] = err;
    cblk[(int64_t)slot * nblocks + b] = make_uint4(blk[0], blk[1], blk[2], blk[3]);
}

// Ordered argmin over the parts of one mode's split scan.
template <int MODE>
__device__ __forceinline__ void merge_parts(Win& w, Lane& ln, const uint4* __restrict__ wins, WideDims dims, int parts, int32_t b, int channels)
{
    load_win_wide(w, wins, dims, wide_win_slot(MODE), 0, b);
    for (int p = 1; p < parts; p++) {
        Win x;
        load_win_wide(x, wins, dims, wide_win_slot(MODE), p, b);
        if (x.err < w.err) { w = x; continue; }
        if (x.err != w.err || x.err == ERR_MAX) continue;
        if (MODE == 0 || MODE == 2) {                     // table order: lowest shape index
            if (x.shape < w.shape) w = x;
        } else {                                          // PCA-ranked list order: lowest rank key
            if (MODE == 7) {
                if (channels == 4) { w.key = merge_key<4, 4>(w, ln.tx, ln.T); x.key = merge_key<4, 4>(x, ln.tx, ln.T); }
                else               { w.key = merge_key<4, 3>(w, ln.tx, ln.T); x.key = merge_key<4, 3>(x, ln.tx, ln.T); }
            } else { w.key = merge_key<3, 3>(w, ln.tx, ln.T); x.key = merge_key<3, 3>(x, ln.tx, ln.T); }
            if (x.key < w.key) w = x;
        }
    }
}

#ifndef WIDE_SCAN_WAVES
#define WIDE_SCAN_WAVES 4
#endif
// a scan is split until its waves fill 1 / WIDE_FILL_DIVISOR of the chip's wave slots (tuning constant, like the register budgets)
#ifndef WIDE_FILL_DIVISOR
#define WIDE_FILL_DIVISOR 2
#endif
// RANKED_LISTS: some family scans a PCA-ranked prefix (fast presets): that code keeps 16 sorted keys in registers and runs
// at 3 waves per SIMD like its deep counterpart; the whole-table instantiation (slow presets) compiles it out.
template <bool VEC16, bool RANKED_LISTS>
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(RANKED_LISTS ? 3 : WIDE_SCAN_WAVES, RANKED_LISTS ? 3 : WIDE_SCAN_WAVES)))
bc7_wide_phase1(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, uint4* __restrict__ wins,
                int32_t* __restrict__ cerr, uint4* __restrict__ cblk, const bc7_enc_settings S, const WideTasks tasks, const int ranked13, const int ranked7,
                const int pstride)
{
    const WideDims dims{nblocks, pstride};
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    __shared__ uint2 s_pal[((RANKED_LISTS && ITW_BC7_LANE_PAL) ? 16 : 12) * TPB];
    Lane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB);
    __syncthreads();
    const int32_t gid = blockIdx.x * TPB + threadIdx.x;
    const bool live = gid < nblocks;
    const int32_t b = live ? gid : nblocks - 1;
    ln.keys = nullptr;
    ln.pal = s_pal + threadIdx.x;
    load_block<VEC16>(ln.tx, src, stride, blocks_x, b);
    ln.best_err = ERR_MAX; ln.opaque_err = 0; ln.improved = false;
    ln.best[0] = ln.best[1] = ln.best[2] = ln.best[3] = 0u;

    const int kind = tasks.kind[blockIdx.y], part = tasks.part[blockIdx.y], parts = tasks.parts[blockIdx.y];   // wave-uniform
    if (kind == WK_SCAN02) {
        Win w0, w2;
        search_02(ln, S, w0, w2, part, parts);
        if (live) { store_win_wide(wins, dims, wide_win_slot(0), part, b, w0); if (!S.skip_mode2) store_win_wide(wins, dims, wide_win_slot(2), part, b, w2); }
    } else if (kind == WK_SCAN13) {
        Win w1, w3;
        if (RANKED_LISTS && ranked13) search_two_subset<false, 3, 2>(ln, S, w1, w3, part, parts); else search_two_subset<false, 3, 0>(ln, S, w1, w3, part, parts);
        if (live) { store_win_wide(wins, dims, wide_win_slot(1), part, b, w1); store_win_wide(wins, dims, wide_win_slot(3), part, b, w3); }
    } else {
        Win w7, unused;
        if (S.channels == 4) { if (RANKED_LISTS && ranked7) search_two_subset<true, 4, 2>(ln, S, w7, unused, part, parts); else search_two_subset<true, 4, 0>(ln, S, w7, unused, part, parts); }
        else                 { if (RANKED_LISTS && ranked7) search_two_subset<true, 3, 2>(ln, S, w7, unused, part, parts); else search_two_subset<true, 3, 0>(ln, S, w7, unused, part, parts); }
        if (live) store_win_wide(wins, dims, wide_win_slot(7), part, b, w7);
    }
}

// SINGLES = false: the refine tasks (modes 0,2,1,3,7; need phase 1).  SINGLES = true: the single-subset tasks (mode 4/5
// candidates, mode 6), which depend on nothing and run on a second stream beside phase 1 when the caller provides one.
template <bool VEC16, bool SINGLES>
// SINGLES (one mode 4/5 candidate group or mode 6 per wave, nothing else live): 168 VGPRs hold it with 10 cold spills, and 44 KiB of LDS let
// three workgroups share a CU: 3 waves per SIMD run modes 4/5 of a call 14 % faster than 2 (profiles/r06_modes45_mapping_probe.txt)
#ifndef WIDE_SINGLES_WAVES
#define WIDE_SINGLES_WAVES 3
#endif
__global__ void __launch_bounds__(TPB) __attribute__((amdgpu_waves_per_eu(SINGLES ? WIDE_SINGLES_WAVES : 2, SINGLES ? WIDE_SINGLES_WAVES : 2)))
bc7_wide_phase2(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, const uint4* __restrict__ wins,
                int32_t* __restrict__ cerr, uint4* __restrict__ cblk, const bc7_enc_settings S, const WideModes modes, const int pstride)
{
    const WideDims dims{nblocks, pstride};
    __shared__ unsigned short s_seed16[2048];
    __shared__ uint32_t s_seed32[2048];
    __shared__ uint2 s_pal[SINGLES ? (ITW_BC7_LANE_PAL ? 16 : 8) * TPB : (ITW_BC7_LANE_PAL ? LANE_PAL_LEVELS * TPB : 1)];   // mode 4/5 vector part, mode 6 (16 levels) / refinement palettes
    Lane ln;
    ln.T = stage_seed_tables_fast(s_seed16, s_seed32, threadIdx.x, TPB);
    __syncthreads();
    const int32_t gid = blockIdx.x * TPB + threadIdx.x;
    const bool live = gid < nblocks;
    const int32_t b = live ? gid : nblocks - 1;
    ln.keys = nullptr;
    ln.pal = s_pal + ((SINGLES || ITW_BC7_LANE_PAL) ? threadIdx.x : 0);
    load_block<VEC16>(ln.tx, src, stride, blocks_x, b);
    ln.best_err = ERR_MAX; ln.improved = false;
    ln.best[0] = ln.best[1] = ln.best[2] = ln.best[3] = 0u;
    ln.opaque_err = 0;                                                             // kernel.ispc:1267-1277
    if (S.channels == 4) {
        uint32_t e = 0;
#pragma unroll
        for (int d = 0; d < 4; d++) { const uint32_t x = ~ln.tx.pl[3][d]; e = udot4(x, x, e); }
        ln.opaque_err = (int32_t)e;
    }
    const int mode = modes.mode[blockIdx.y], parts = modes.parts[blockIdx.y];      // wave-uniform
    if (SINGLES && mode >= 100) {                                                  // one mode 4/5 candidate of one rotation
        const int r = (mode - 100) >> 2, c = (mode - 100) & 3;
        Dual best4, best5;
        clear(best4); clear(best5);
        int32_t err4 = ERR_MAX, err5 = ERR_MAX;
        modes_45_scan(ln, S, r, r + 1, best4, err4, best5, err5, c == 3 ? 7 : 1 << c);       // c == 3: all three in this wave
        uint32_t blk[4] = {0u, 0u, 0u, 0u};
        if (c != 2) {                                     // mode 4 (c == 3: the better of both swaps, first wins a tie: slot of swap 0)
            if (err4 < ERR_MAX) emit_dual<4>(blk, best4);
            if (live) store_candidate(cerr, cblk, nblocks, WIDE_SLOT_M4 + 2 * r + (c == 1 ? 1 : 0), b, err4, blk);
        }
        if (c >= 2) {
            blk[0] = blk[1] = blk[2] = blk[3] = 0u;
            if (err5 < ERR_MAX) emit_dual<5>(blk, best5);
            if (live) store_candidate(cerr, cblk, nblocks, WIDE_SLOT_M5 + r, b, err5, blk);
        }
        return;
    }
    if (SINGLES) {                                                                 // mode 6
        if (S.channels == 4) mode_6<4>(ln, S); else mode_6<3>(ln, S);
        if (live) store_candidate(cerr, cblk, nblocks, WIDE_SLOT_M6, b, ln.best_err, ln.best);
        return;
    }
    Win w;
    int slot = 0;
    if (mode == 0)      { merge_parts<0>(w, ln, wins, dims, parts, b, S.channels); refine_and_commit<0>(ln, w, S.refineIterations[0], S.channels); slot = 0; }
    else if (mode == 2) { merge_parts<2>(w, ln, wins, dims, parts, b, S.channels); refine_and_commit<2>(ln, w, S.refineIterations[2], S.channels); slot = 1; }
    else if (mode == 1) { merge_parts<1>(w, ln, wins, dims, parts, b, S.channels); refine_and_commit<1>(ln, w, S.refineIterations[1], S.channels); slot = 2; }
    else if (mode == 3) { merge_parts<3>(w, ln, wins, dims, parts, b, S.channels); refine_and_commit<3>(ln, w, S.refineIterations[3], S.channels); slot = 3; }
    else                { merge_parts<7>(w, ln, wins, dims, parts, b, S.channels); refine_and_commit<7>(ln, w, S.refineIterations[7], S.channels); slot = 4; }
    if (live) store_candidate(cerr, cblk, nblocks, slot, b, ln.best_err, ln.best);
}

__global__ void __launch_bounds__(TPB)
bc7_wide_commit(const int32_t* __restrict__ cerr, const uint4* __restrict__ cblk, int32_t nblocks, uint32_t active_slots, uint4* __restrict__ dst, int vec16)
{
    const int32_t b = blockIdx.x * TPB + threadIdx.x;
    if (b >= nblocks) return;
    int32_t best = ERR_MAX;
    uint4 blk = make_uint4(0u, 0u, 0u, 0u);
    for (int slot = 0; slot < WIDE_SLOTS; slot++) {
        if (!((active_slots >> slot) & 1u)) continue;
        const int32_t e = cerr[(int64_t)slot * nblocks + b];
        if (e < best) { best = e; blk = cblk[(int64_t)slot * nblocks + b]; }
    }
    if (vec16) dst[b] = blk;
    else { uint32_t* d = reinterpret_cast<uint32_t*>(dst) + (int64_t)b * 4; d[0] = blk.x; d[1] = blk.y; d[2] = blk.z; d[3] = blk.w; }
}

struct Bc7Launch {
    bool vec; dim3 grid; hipStream_t st; const uint8_t* src; int64_t stride; int bx; int32_t n;
    uint8_t* dst; int32_t* err; uint4* wins; bc7_enc_settings S; int first;
};

template <int FAMILY, int RANKED>
static void launch_search(const Bc7Launch& L)
{
    const size_t lds = RANKED == 1 ? (size_t)64 * TPB * sizeof(int32_t) : 0;
    if (L.vec) hipLaunchKernelGGL((bc7_search_kernel<FAMILY, RANKED, true>),  L.grid, dim3(TPB), lds, L.st, L.src, L.stride, L.bx, L.n, L.wins, L.S);
    else       hipLaunchKernelGGL((bc7_search_kernel<FAMILY, RANKED, false>), L.grid, dim3(TPB), lds, L.st, L.src, L.stride, L.bx, L.n, L.wins, L.S);
}
template <int FAMILY>
static void launch_finish(Bc7Launch& L)
{
    if (L.vec) hipLaunchKernelGGL((bc7_finish_kernel<FAMILY, true>),  L.grid, dim3(TPB), 0, L.st, L.src, L.stride, L.bx, L.n, L.dst, L.err, L.wins, L.S, L.first);
    else       hipLaunchKernelGGL((bc7_finish_kernel<FAMILY, false>), L.grid, dim3(TPB), 0, L.st, L.src, L.stride, L.bx, L.n, L.dst, L.err, L.wins, L.S, L.first);
    L.first = 0;
}

// Calls of at most this many blocks take the wide path (measured crossover on MI355X, tools/bc7_path_probe.py).
#ifndef ITW_BC7_WIDE_MAX_BLOCKS
#define ITW_BC7_WIDE_MAX_BLOCKS 262144
#endif

// 0 = by size, 1 = deep always, 2 = wide whenever it supports the settings (itwSetBc7Path / ITW_BC7_PATH=deep|wide:
// tests and probes)
#ifdef ITW_TEST_HOOKS
// test hook (itwTestBc7TwoSubsetBounds, libispc_texcomp_test.so only): two_subset_bound of every two-subset shape of every block, out[block * 64 + shape]
template <bool VEC16>
__global__ void __launch_bounds__(TPB) bc7_test_bounds(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, float* __restrict__ out)
{
    const int32_t gid = blockIdx.x * TPB + threadIdx.x;
    const int32_t b = gid < nblocks ? gid : nblocks - 1;
    Tex tx;
    load_block<VEC16>(tx, src, stride, blocks_x, b);
    IStats<3> full;
    stats_int<3>(full, tx.pl, whole_block());
#pragma unroll 1
    for (int shape = 0; shape < 64; shape++) {
        const float lb = two_subset_bound(shape, tx.pl, full);
        if (gid < nblocks) out[(int64_t)b * 64 + shape] = lb;
    }
}
void launch_bc7_test_bounds(const uint8_t* src, int64_t stride, int width, int height, float* out, hipStream_t st)
{
    const int bx = width / 4;
    const int64_t n = (int64_t)bx * (height / 4);
    if (n <= 0) return;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 15) == 0;
    const dim3 grid((unsigned)((n + TPB - 1) / TPB));
    if (vec) hipLaunchKernelGGL((bc7_test_bounds<true>),  grid, dim3(TPB), 0, st, src, stride, bx, (int32_t)n, out);
    else     hipLaunchKernelGGL((bc7_test_bounds<false>), grid, dim3(TPB), 0, st, src, stride, bx, (int32_t)n, out);
}
#endif // ITW_TEST_HOOKS

static std::atomic<int> g_bc7_path{-1};
static int bc7_path_override()
{
    int v = g_bc7_path.load(std::memory_order_relaxed);
    if (v < 0) {
        const char* e = std::getenv("ITW_BC7_PATH");
        v = !e ? 0 : !std::strcmp(e, "deep") ? 1 : !std::strcmp(e, "wide") ? 2 : 0;
        g_bc7_path.store(v, std::memory_order_relaxed);
    }
    return v;
}
void set_bc7_path(int v) { g_bc7_path.store(v == 1 || v == 2 ? v : 0, std::memory_order_relaxed); }
bool bc7_staged_bands_ok()
{
    static const bool on = [] { const char* e = std::getenv("ITW_STAGED_BANDS"); return !(e && e[0] == '0'); }();
    return on && bc7_path_override() != 2;
}

// RGBA profile with both groups of modes: the fused shape runs the alpha-capable modes first and skips the three-channel
// modes where they cannot win (bc7_finish_all).  ITW_BC7_ALPHA_PRUNE=0: always the reference's order.
// ITW_BC7_BOUND=0: modes 1/3 in the reference's place instead of last-and-bounded (same bytes; tools/gpu_env_matrix.sh runs both)
static bool bc7_bounded_order()
{
    static const bool on = [] { const char* e = std::getenv("ITW_BC7_BOUND"); return !(e && e[0] == '0'); }();
    return on;
}
// Round 5.  ITW_BC7_PILOT_THR: the pilot's threshold in percent of the blocks it looks at that its estimate lists for modes 1/3 -- at or below it the rest of
// the surface takes the bounded order, above it the reference's; -1 = no pilot (always bounded), 0 = pilot, always the reference's order for the
// rest, 100 = pilot, always bounded (tools/order_timing.py under each setting: profiles/history/r05/r05e_*).  Returned in 1/256.
#ifndef ITW_BC7_PILOT_THR_DEFAULT
#define ITW_BC7_PILOT_THR_DEFAULT 90     // forced either way by content (profiles/history/r05/r05l_*): 74 % estimated -> bounded wins by 11 %, 91 % -> equal, 98 % -> reference order by 0.6 %
#endif
static std::atomic<int> g_bc7_pilot{-2};           // percent; -2 = not read yet
static int bc7_pilot_threshold()
{
    int pct = g_bc7_pilot.load(std::memory_order_relaxed);
    if (pct == -2) {
        const char* e = std::getenv("ITW_BC7_PILOT_THR");
        pct = e ? std::atoi(e) : ITW_BC7_PILOT_THR_DEFAULT;
        pct = pct < 0 ? -1 : (pct > 100 ? 100 : pct);
        g_bc7_pilot.store(pct, std::memory_order_relaxed);
    }
    return pct < 0 ? -1 : pct * 256 / 100;
}
void set_bc7_pilot(int percent) { g_bc7_pilot.store(percent < -1 ? -2 : (percent > 100 ? 100 : percent), std::memory_order_relaxed); }   // < -1: back to ITW_BC7_PILOT_THR / the default
static bool bc7_pilot_debug()
{
    static const bool on = [] { const char* e = std::getenv("ITW_BC7_PILOT_DEBUG"); return e && e[0] == '1'; }();
    return on;
}
// ITW_BC7_BANDS=1: one band on the caller's stream, no pilot (round 4's launch order); default 2
static int bc7_bands()
{
    static const int k = [] { const char* e = std::getenv("ITW_BC7_BANDS"); const int v = e ? std::atoi(e) : 2; return v < 2 ? 1 : 2; }();
    return k;
}
// ITW_BC7_COMPACT=0: the list scans gather their blocks from the surface (round 4) instead of reading the compact copy
static bool bc7_compact_lists()
{
    static const bool on = [] { const char* e = std::getenv("ITW_BC7_COMPACT"); return !(e && e[0] == '0'); }();
    return on;
}
static bool bc7_fused_on()
{
    static const bool on = [] { const char* e = std::getenv("ITW_BC7_FUSED"); return !(e && e[0] == '0'); }();
    return on;
}
static bool bc7_alpha_first(const bc7_enc_settings& S)
{
    static const bool on = [] { const char* e = std::getenv("ITW_BC7_ALPHA_PRUNE"); return !(e && e[0] == '0'); }();
    const bool on02 = S.mode_selection[0];
    const bool on13 = S.mode_selection[1] && (S.fastSkipTreshold_mode1 > 0 || S.fastSkipTreshold_mode3 > 0);
    const bool on7 = S.mode_selection[1] && S.fastSkipTreshold_mode7 > 0;
    return on && S.channels == 4 && (on02 || on13) && (on7 || S.mode_selection[2] || S.mode_selection[3]);
}
// Above this many blocks an alpha-first call takes the fused shape even where the wide one is allowed: on translucent
// content it is 1.3-2.3x faster from 65536 blocks up, on opaque content (nothing to skip) at most 8 % slower from here
// (tools/bc7_path_probe.py, profiles/r02c_bc7_path_probe.txt vs r02b: alpha_basic 131072 blocks 0.361 / 0.758 vs 0.681 ms wide)
#define ITW_BC7_WIDE_MAX_BLOCKS_ALPHA_FIRST 131072

static bool bc7_use_wide(int64_t n, const bc7_enc_settings& S, int64_t wide_max)
{
    // lists longer than 16 that are proper prefixes of the ranking keep their keys in 64 KiB of LDS: deep path only
    auto long_ranked = [](int t) { return t > 16 && t < 64; };
    if (S.mode_selection[1] && (long_ranked(S.fastSkipTreshold_mode1) || long_ranked(S.fastSkipTreshold_mode3) || long_ranked(S.fastSkipTreshold_mode7))) return false;
    // the wide ranked scan keeps the 16 smallest keys in registers: a family with a ranked list needs all its lists <= 16
    auto ranked = [](int t) { return t > 0 && t < 64; };
    if (S.mode_selection[1]) {
        const int a = S.fastSkipTreshold_mode1, c = S.fastSkipTreshold_mode3;
        if ((ranked(a) || ranked(c)) && (a > 16 || c > 16)) return false;
    }
    if (S.mode45_channel0 < 0 || S.mode45_channel0 > 3) return false;
    const int o = bc7_path_override();
    if (o == 1) return false;
    const int64_t hard_cap = (int64_t)1 << 20;                   // workspace bound of the wide layout (1.6 GB)
    if (o == 2) return n <= hard_cap;
    if (n >= ITW_BC7_WIDE_MAX_BLOCKS_ALPHA_FIRST && bc7_alpha_first(S)) return false;
    return n <= (wide_max > 0 ? (wide_max < hard_cap ? wide_max : hard_cap) : ITW_BC7_WIDE_MAX_BLOCKS);
}

// deep: best error so far (4 B/block) + the winners of one family's two modes (2 x 16 B/block)
// wide: winners [5 modes][parts] x 16 B + candidates [18 slots] x (4 + 16) B per block
// a scan is split while that keeps all its waves resident at once (launch_bc7_wide): parts * blocks <= max(blocks, 262144)
static size_t wide_win_entries(size_t n) { return n > 262144 ? n : 262144; }
static size_t wide_workspace_bytes(size_t n)
{
    return (size_t)5 * wide_win_entries(n) * sizeof(uint4) + (((size_t)WIDE_SLOTS * n * sizeof(int32_t) + 15) & ~(size_t)15) + (size_t)WIDE_SLOTS * n * sizeof(uint4);
}
// fused shape, in 4-byte words from the start of the workspace (every region starts on a 16-byte boundary)
// a list of blocks for modes 1/3 with what its split scan and refinement need: block ids [cap], winners of the scan's shares [5 rows][cap]
// (rows 2, 3 used: list_scan_parts x listed <= cap), the listed blocks' texels [cap] x 64 B in list order
struct ListRegion { size_t list, wins, compact, cap; };
struct FusedLayout { size_t inc, list0, list1, list2, counts, words; ListRegion band[2]; };
// Which regions a call's settings can touch (a `basic` call was sized like a `slow` one, 124 B per block instead of 36):
//   lists    the three block lists of the RGBA profiles' alpha-first order (and list13 of an RGB bounded call that runs as one band);
//   bands    the per-band lists and share winners of the bounded order (modes 1/3, mode 7 scanned over lists);
//   compact  the listed blocks' texels in list order (the RGB bounded order only).
struct FusedNeeds { bool lists, bands, compact; };
static FusedNeeds fused_needs(const bc7_enc_settings* s)
{
    if (!s) return FusedNeeds{true, true, true};                  // unknown settings: everything
    bc7_enc_settings S = *s;
    S.channels = (s->channels == 4) ? 4 : 3;
    auto ranked = [](int t) { return t > 0 && t < 64; };
    const bool on02 = S.mode_selection[0];
    const bool on13 = S.mode_selection[1] && (S.fastSkipTreshold_mode1 > 0 || S.fastSkipTreshold_mode3 > 0);
    const bool r13 = on13 && (ranked(S.fastSkipTreshold_mode1) || ranked(S.fastSkipTreshold_mode3));
    const bool bounded = bc7_bounded_order() && on02 && on13 && !r13;
    const bool alpha = bc7_alpha_first(S);
    return FusedNeeds{alpha || bounded, bounded, bounded && !alpha && bc7_compact_lists()};
}
static FusedLayout fused_layout(size_t n, const FusedNeeds& need)
{
    auto up4 = [](size_t w) { return (w + 3) & ~(size_t)3; };
    const size_t nchunks = (n + TPB - 1) / TPB;
    FusedLayout W;
    size_t o = up4(5 * n);                                         // 5 winner rows
    W.inc = o;      o = up4(o + n);                                // a phase's error / incumbent
    const size_t lcap = need.lists ? 2 * (nchunks / 2 + 65) * TPB : 0;   // three block lists (RGBA profiles), each with room for two bands' shares
    W.list0 = o;    o = up4(o + lcap);
    W.list1 = o;    o = up4(o + lcap);
    W.list2 = o;    o = up4(o + lcap);
    W.counts = o;   o += 16;                                       // list lengths, the pilot's verdict
    for (int k = 0; k < 2; k++) {                                  // the two bands: lists and share winners ...
        ListRegion& r = W.band[k];
        r.cap = (nchunks / 2 + 65) * TPB;                          // (a band is whole stripes of up to 64 chunks)
        r.list = o;    o = up4(o + (need.bands ? r.cap : 0));
        r.wins = o;    o = up4(o + (need.bands ? 5 * r.cap : 0));
    }
    W.band[0].compact = o; o += need.compact ? 16 * W.band[0].cap : 0;   // ... and their texels, contiguous: a call without bands uses both as one
    W.band[1].compact = o; o += need.compact ? 16 * W.band[1].cap : 0;
    W.words = o;
    return W;
}
bool bc7_scans_every_shape(const bc7_enc_settings& s) { return fused_needs(&s).bands; }

size_t bc7_workspace_bytes(int width, int height, int64_t wide_max_blocks, const bc7_enc_settings* settings)
{
    const size_t n = (size_t)(width / 4) * (size_t)(height / 4);
    size_t deep = ((n * sizeof(int32_t) + 15) & ~(size_t)15) + 2 * n * sizeof(uint4);
    const size_t fused = fused_layout(n, fused_needs(settings)).words * sizeof(uint32_t);
    if (fused > deep) deep = fused;
    const size_t limit = bc7_path_override() == 2 ? ((size_t)1 << 20) : (wide_max_blocks > 0 ? (size_t)wide_max_blocks : (size_t)ITW_BC7_WIDE_MAX_BLOCKS);
    const bool may_wide = bc7_path_override() != 1 && n <= limit && n <= ((size_t)1 << 20);
    const size_t wide = may_wide ? wide_workspace_bytes(n) : 0;
    return deep > wide ? deep : wide;
}

static void launch_bc7_wide(const uint8_t* src, int64_t stride, int bx, int64_t n, uint8_t* dst, const bc7_enc_settings& S, float* workspace,
                            hipStream_t st, const Bc7Aux* aux)
{
    uint4* wins = reinterpret_cast<uint4*>(workspace);
    int32_t* cerr = reinterpret_cast<int32_t*>(wins + (size_t)5 * wide_win_entries((size_t)n));
    uint4* cblk = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(cerr) + (((size_t)WIDE_SLOTS * n * sizeof(int32_t) + 15) & ~(size_t)15));
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    auto ranked = [](int t) { return t > 0 && t < 64; };

    const bool on02 = S.mode_selection[0];
    const bool on13 = S.mode_selection[1] && (S.fastSkipTreshold_mode1 > 0 || S.fastSkipTreshold_mode3 > 0);
    const bool on7  = S.mode_selection[1] && S.fastSkipTreshold_mode7 > 0;
    const bool on45 = S.mode_selection[2], on6 = S.mode_selection[3];
    const int ranked13 = on13 && (ranked(S.fastSkipTreshold_mode1) || ranked(S.fastSkipTreshold_mode3));
    const int ranked7 = on7 && ranked(S.fastSkipTreshold_mode7);
    const bool any_ranked = ranked13 || ranked7;

    // parts per scan: as many waves as stay resident at once (1024 SIMDs x the scan kernel's waves), within what the
    // candidate lists offer
    const int64_t waves = (n + 63) / 64;
    const int families = (on02 ? 1 : 0) + (on13 ? 1 : 0) + (on7 ? 1 : 0);
    const int resident = 1024 * (any_ranked ? 3 : WIDE_SCAN_WAVES);
    int want = 1;
    while (want < WIDE_MAX_PARTS && waves * families * want * WIDE_FILL_DIVISOR <= resident) want *= 2;
    auto cap = [&](int limit) { int p = want; while (p > limit) p /= 2; return p < 1 ? 1 : p; };
    const int p02 = cap(16);                                                        // 16 or 64 shapes
    const int list13 = ranked13 ? (S.fastSkipTreshold_mode1 > S.fastSkipTreshold_mode3 ? S.fastSkipTreshold_mode1 : S.fastSkipTreshold_mode3) : 64;
    const int p13 = cap(ranked13 ? (list13 >= 8 ? 4 : list13 >= 2 ? 2 : 1) : 16);   // a ranked share recomputes the 64 keys: few parts
    const int list7 = ranked7 ? S.fastSkipTreshold_mode7 : 64;
    const int p7 = cap(ranked7 ? (list7 >= 8 ? 4 : list7 >= 2 ? 2 : 1) : 16);
    int pstride = 1;                                                               // rows of the winners array per mode
    if (on02 && p02 > pstride) pstride = p02;
    if (on13 && p13 > pstride) pstride = p13;
    if (on7 && p7 > pstride) pstride = p7;

    WideTasks T;
    std::memset(&T, 0, sizeof T);
    auto add = [&](int kind, int part, int parts) { T.kind[T.n] = (uint8_t)kind; T.part[T.n] = (uint8_t)part; T.parts[T.n] = (uint8_t)parts; T.n++; };
    WideModes R, G;                                       // refine tasks (after the scans), single-subset tasks (independent)
    std::memset(&R, 0, sizeof R);
    std::memset(&G, 0, sizeof G);
    uint32_t active = 0;
    auto add_refine = [&](int mode, int parts) { R.mode[R.n] = (uint8_t)mode; R.parts[R.n] = (uint8_t)parts; R.n++; active |= 1u << wide_win_slot(mode); };
    auto add_single = [&](int code) { G.mode[G.n] = (uint8_t)code; G.parts[G.n] = 1; G.n++; };
    // longest tasks first: a launch drains in blockIdx order
    if (on13) { for (int p = 0; p < p13; p++) add(WK_SCAN13, p, p13); if (S.fastSkipTreshold_mode1 > 0) add_refine(1, p13); if (S.fastSkipTreshold_mode3 > 0) add_refine(3, p13); }
    if (on02) { for (int p = 0; p < p02; p++) add(WK_SCAN02, p, p02); add_refine(0, p02); if (!S.skip_mode2) add_refine(2, p02); }
    if (on7)  { for (int p = 0; p < p7; p++) add(WK_SCAN7, p, p7); add_refine(7, p7); }
    if (on45) {
        // one wave per rotation, or one per (rotation, candidate) while all those waves (2 per SIMD) stay resident at once
        const int rots = (S.channels == 4 ? 4 : 3) - S.mode45_channel0;
        const bool split = waves * (3 * rots + (on6 ? 1 : 0) + (aux ? 0 : R.n)) <= 2048;
        for (int r = S.mode45_channel0; r < (S.channels == 4 ? 4 : 3); r++) {
            if (split) { for (int c = 0; c < 3; c++) add_single(100 + 4 * r + c); active |= 3u << (WIDE_SLOT_M4 + 2 * r); }
            else       { add_single(100 + 4 * r + 3); active |= 1u << (WIDE_SLOT_M4 + 2 * r); }
            active |= 1u << (WIDE_SLOT_M5 + r);
        }
    }
    if (on6)  { add_single(6); active |= 1u << WIDE_SLOT_M6; }

    const unsigned gx = (unsigned)((n + TPB - 1) / TPB);
    const dim3 blk(TPB);
    // the single-subset tasks need nothing from the scans: beside them on the caller-provided second stream, else after them
    hipStream_t gs = st;
    if (aux && G.n > 0 && T.n > 0) {
        (void)hipEventRecord(aux->fork, st);               // whatever feeds `src` on st (an upload) comes first
        (void)hipStreamWaitEvent(aux->stream, aux->fork, 0);
        gs = aux->stream;
    }
    if (G.n > 0) {
        if (vec) hipLaunchKernelGGL((bc7_wide_phase2<true, true>),  dim3(gx, (unsigned)G.n), blk, 0, gs, src, stride, bx, (int32_t)n, wins, cerr, cblk, S, G, pstride);
        else     hipLaunchKernelGGL((bc7_wide_phase2<false, true>), dim3(gx, (unsigned)G.n), blk, 0, gs, src, stride, bx, (int32_t)n, wins, cerr, cblk, S, G, pstride);
        if (gs != st) (void)hipEventRecord(aux->join, gs);
    }
    if (T.n > 0) {
        const dim3 grid(gx, (unsigned)T.n);
        if (any_ranked) {
            if (vec) hipLaunchKernelGGL((bc7_wide_phase1<true, true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, S, T, ranked13, ranked7, pstride);
            else     hipLaunchKernelGGL((bc7_wide_phase1<false, true>), grid, blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, S, T, ranked13, ranked7, pstride);
        } else {
            if (vec) hipLaunchKernelGGL((bc7_wide_phase1<true, false>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, S, T, ranked13, ranked7, pstride);
            else     hipLaunchKernelGGL((bc7_wide_phase1<false, false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, S, T, ranked13, ranked7, pstride);
        }
    }
    if (R.n > 0) {
        if (vec) hipLaunchKernelGGL((bc7_wide_phase2<true, false>),  dim3(gx, (unsigned)R.n), blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, S, R, pstride);
        else     hipLaunchKernelGGL((bc7_wide_phase2<false, false>), dim3(gx, (unsigned)R.n), blk, 0, st, src, stride, bx, (int32_t)n, wins, cerr, cblk, S, R, pstride);
    }
    if (gs != st) (void)hipStreamWaitEvent(st, aux->join, 0);
    hipLaunchKernelGGL(bc7_wide_commit, dim3(gx), blk, 0, st, cerr, cblk, (int32_t)n, active, reinterpret_cast<uint4*>(dst), vec ? 1 : 0);
}

// does a call with these settings run the bounded order of the RGB profiles (`slow`: the one order with a pilot)?
bool bc7_has_order_verdict(const bc7_enc_settings& s)
{
    bc7_enc_settings S = s;
    S.channels = (s.channels == 4) ? 4 : 3;
    auto ranked = [](int t) { return t > 0 && t < 64; };
    const bool on13 = S.mode_selection[1] && (S.fastSkipTreshold_mode1 > 0 || S.fastSkipTreshold_mode3 > 0);
    return bc7_fused_on() && bc7_bounded_order() && S.mode_selection[0] && on13 && !ranked(S.fastSkipTreshold_mode1) && !ranked(S.fastSkipTreshold_mode3) &&
           !(S.mode_selection[1] && S.fastSkipTreshold_mode7 > 0) && !bc7_alpha_first(S);
}

// Families run in the reference's order (kernel.ispc:1970-1977): {0,2} -> {1,3} -> {7} -> {4,5} -> {6}.
void launch_bc7(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst,
                const bc7_enc_settings& s, float* workspace, hipStream_t st, const Bc7Aux* aux)
{
    const int bx = width / 4, by = height / 4;
    const int64_t n = (int64_t)bx * by;
    if (n <= 0) return;
    Bc7Launch L;
    L.S = s;
    L.S.channels = (s.channels == 4) ? 4 : 3;
    const bool single = aux && aux->single;
    if (aux && aux->probe && !bc7_has_order_verdict(L.S)) return;   // only the bounded order of the RGB profiles has a verdict to estimate
    if (!single && bc7_use_wide(n, L.S, aux ? aux->wide_max_blocks : 0)) { launch_bc7_wide(src, stride, bx, n, dst, L.S, workspace, st, aux); return; }
    L.err = reinterpret_cast<int32_t*>(workspace);
    L.wins = reinterpret_cast<uint4*>(reinterpret_cast<uint8_t*>(workspace) + (((size_t)n * sizeof(int32_t) + 15) & ~(size_t)15));
    L.vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    L.grid = dim3((unsigned)((n + TPB - 1) / TPB));
    L.st = st; L.src = src; L.stride = stride; L.bx = bx; L.n = (int32_t)n; L.dst = dst; L.first = 1;
    const bc7_enc_settings& S = L.S;
    auto ranked = [](int t) { return t > 0 && t < 64; };
    auto small = [](int t) { return t <= 16; };                  // non-positive thresholds disable the mode
    {
        // two launches (scan of all families, refinement of all modes) unless a ranked list is longer than 16 shapes (those
        // keep their keys in 64 KiB of LDS: the per-family kernels below) or ITW_BC7_FUSED=0
        const bool fused_on = bc7_fused_on();
        const bool on02 = S.mode_selection[0];
        const bool on13 = S.mode_selection[1] && (S.fastSkipTreshold_mode1 > 0 || S.fastSkipTreshold_mode3 > 0);
        const bool on7 = S.mode_selection[1] && S.fastSkipTreshold_mode7 > 0;
        const bool r13 = on13 && (ranked(S.fastSkipTreshold_mode1) || ranked(S.fastSkipTreshold_mode3));
        const bool r7 = on7 && ranked(S.fastSkipTreshold_mode7);
        const bool long13 = r13 && !(small(S.fastSkipTreshold_mode1) && small(S.fastSkipTreshold_mode3));
        const bool long7 = r7 && !small(S.fastSkipTreshold_mode7);
        const bool any_mode = on02 || on13 || on7 || S.mode_selection[2] || S.mode_selection[3];
        if (fused_on && !long13 && !long7 && any_mode) {
            uint32_t* wins4 = reinterpret_cast<uint32_t*>(workspace);          // [5 modes][n] x 4 B
            const int32_t nchunks = (int32_t)((n + TPB - 1) / TPB);
            // Interleave grain, measured at 4096^2 `slow` (scan time / scan FETCH_SIZE): 8 chunks 6.01 ms / 76 MB (two code paths
            // alternate on every CU: instruction cache), 256 chunks 5.36 ms / 76 MB, 1024 chunks 5.38 ms / 166 MB (L2 no longer
            // holds the first family's texels), family-major 5.39 ms / 154 MB; two separate launches 5.45 ms.
            const int a13 = r13 ? 1 : 0, a7 = r7 ? 1 : 0;
            const FusedLayout W = fused_layout((size_t)n, fused_needs(&S));
            int32_t* alpha_err = reinterpret_cast<int32_t*>(wins4 + W.inc);                    // [n] x 4 B behind the winner rows: a phase's error / incumbent
            int32_t* rgb_list = reinterpret_cast<int32_t*>(wins4 + W.list0);                   // [n] block ids (RGBA profiles: where an RGB mode can win; RGB: where modes 1/3 can)
            int32_t* list13 = reinterpret_cast<int32_t*>(wins4 + W.list1);                     // [n] block ids (RGBA profiles, bounded order)
            int32_t* list7 = reinterpret_cast<int32_t*>(wins4 + W.list2);                      // [n] block ids (RGBA profiles, bounded order incl. mode 7)
            int32_t* rgb_count = reinterpret_cast<int32_t*>(wins4 + W.counts);                 // the lists' lengths, the pilot's list length and verdict
            int32_t* count13 = rgb_count + 1;
            int32_t* band_count = rgb_count + 3;                                               // [2]: the bands' lists for modes 1/3
            int32_t* pilot_flag = rgb_count + 6;                                               // the pilot's verdict
            int32_t* pilot_ctr = rgb_count + 8;                                                // [3]: its counts (listed, sampled, workgroups done)
            const bool compact_on = bc7_compact_lists();
            const dim3 blk(TPB);
            const ChunkSel ALL{0, 1, 0, nullptr, 0};
            // `sel`: which chunks (ChunkSel); `cnt`: how many chunks that is; `rows`: length of a winner row (n; a band's list scan has its own rows)
            auto scan_rgb = [&](const int32_t* list, const int32_t* count, bool do13, bool do02, int32_t split = 0, ChunkSel sel = ChunkSel{0, 1, 0, nullptr, 0},
                                int32_t cnt = -1, hipStream_t s = nullptr, uint32_t* wins = nullptr, int32_t rows = 0, const uint4* compact = nullptr) {
                ScanTasks T;
                T.n = 0;
                if (on13 && do13) T.kind[T.n++] = WK_SCAN13;                   // longest first
                if (on02 && do02) T.kind[T.n++] = WK_SCAN02;
                if (T.n == 0) return;
                if (cnt < 0) cnt = nchunks;
                if (!s) s = st;
                if (!wins) { wins = wins4; rows = (int32_t)n; }
                const int32_t c8 = ((cnt + 7) / 8) * 8;
                int32_t grain = 256;
                if (grain > c8) grain = c8;
                const int32_t groups = (c8 + grain - 1) / grain;
                const dim3 grid((unsigned)(groups * grain * T.n));
                if (r13) {
                    if (L.vec) hipLaunchKernelGGL((bc7_scan_all<true, true, false>),  grid, blk, 0, s, src, stride, bx, rows, wins, S, T, a13, a7, cnt, grain, list, count, split, sel, compact);
                    else       hipLaunchKernelGGL((bc7_scan_all<false, true, false>), grid, blk, 0, s, src, stride, bx, rows, wins, S, T, a13, a7, cnt, grain, list, count, split, sel, compact);
                } else {
                    if (L.vec) hipLaunchKernelGGL((bc7_scan_all<true, false, false>),  grid, blk, 0, s, src, stride, bx, rows, wins, S, T, a13, a7, cnt, grain, list, count, split, sel, compact);
                    else       hipLaunchKernelGGL((bc7_scan_all<false, false, false>), grid, blk, 0, s, src, stride, bx, rows, wins, S, T, a13, a7, cnt, grain, list, count, split, sel, compact);
                }
            };
            auto scan_7 = [&](const int32_t* list = nullptr, const int32_t* count = nullptr, int32_t split = 0, int32_t cnt = -1, hipStream_t s = nullptr,
                              uint32_t* wins = nullptr, int32_t rows = 0) {
                if (!on7) return;
                ScanTasks T7;
                T7.n = 1; T7.kind[0] = WK_SCAN7;
                if (cnt < 0) cnt = nchunks;
                if (!s) s = st;
                if (!wins) { wins = wins4; rows = (int32_t)n; }
                const int32_t chunks8 = ((cnt + 7) / 8) * 8;
                const dim3 grid((unsigned)chunks8);
                if (r7) {
                    if (L.vec) hipLaunchKernelGGL((bc7_scan_all<true, true, true>),  grid, blk, 0, s, src, stride, bx, rows, wins, S, T7, a13, a7, cnt, chunks8, list, count, split, ALL, (const uint4*)nullptr);
                    else       hipLaunchKernelGGL((bc7_scan_all<false, true, true>), grid, blk, 0, s, src, stride, bx, rows, wins, S, T7, a13, a7, cnt, chunks8, list, count, split, ALL, (const uint4*)nullptr);
                } else {
                    if (L.vec) hipLaunchKernelGGL((bc7_scan_all<true, false, true>),  grid, blk, 0, s, src, stride, bx, rows, wins, S, T7, a13, a7, cnt, chunks8, list, count, split, ALL, (const uint4*)nullptr);
                    else       hipLaunchKernelGGL((bc7_scan_all<false, false, true>), grid, blk, 0, s, src, stride, bx, rows, wins, S, T7, a13, a7, cnt, chunks8, list, count, split, ALL, (const uint4*)nullptr);
                }
            };
            // a finish phase: list phases launch one workgroup per possible list chunk (they return at once behind the list's end)
            auto finish = [&](auto phase, const int32_t* in_list, const int32_t* in_count, int32_t* out_list, int32_t* out_count,
                              int32_t* out_list7 = nullptr, int32_t* out_count7 = nullptr, ChunkSel sel = ChunkSel{0, 1, 0, nullptr, 0}, int32_t cnt = -1,
                              hipStream_t s = nullptr, const uint32_t* wins = nullptr, int32_t rows = 0, const uint4* in_compact = nullptr, uint4* out_compact = nullptr) {
                constexpr int PH = decltype(phase)::value;
                if (cnt < 0) cnt = nchunks;
                if (!s) s = st;
                if (!wins) { wins = wins4; rows = (int32_t)n; }
                const dim3 grid((unsigned)cnt);
                if (L.vec) hipLaunchKernelGGL((bc7_finish_all<true, PH>),  grid, blk, 0, s, src, stride, bx, rows, dst, wins, S, alpha_err, in_list, in_count, out_list, out_count, out_list7, out_count7, sel, in_compact, out_compact);
                else       hipLaunchKernelGGL((bc7_finish_all<false, PH>), grid, blk, 0, s, src, stride, bx, rows, dst, wins, S, alpha_err, in_list, in_count, out_list, out_count, out_list7, out_count7, sel, in_compact, out_compact);
            };
            // the bounded order (bc7_finish_all's header): profiles whose modes 1/3 scan every shape -- a ranked list already spends
            // a bound's worth of arithmetic per shape on its keys and then fits a few shapes only
            const bool bounded = bc7_bounded_order() && on02 && on13 && !r13;
            if (bc7_alpha_first(S)) {
                ITW_CHECK(hipMemsetAsync(rgb_count, 0, 16 * sizeof(int32_t), st));  // the finish phases append to the lists through them
                if (bounded && on7 && !r7 && (S.mode_selection[2] || S.mode_selection[3])) {
                    // mode 7 bounded too: modes 4,5,6 | 0,2 first, then 1,3 and 7 each over its own list.  Round 5: seven dependent launches
                    // leave the chip partly empty seven times, so -- as for the RGB profiles below -- the surface is cut into two interleaved
                    // bands, each with its own lists and share winners, one per stream.
                    const bool two = bc7_bands() > 1 && aux && !aux->single && aux->stream && nchunks >= 32;
                    int32_t stripe = nchunks / 16;
                    stripe = stripe < 1 ? 1 : (stripe > 64 ? 64 : stripe);
                    const int32_t stripes = (nchunks + stripe - 1) / stripe;
                    auto chain = [&](int k, hipStream_t s) {
                        const ChunkSel sel{two ? 1 : 0, stripe, k, nullptr, 0};
                        const int32_t cnt = two ? ((stripes + 1 - k) / 2) * stripe : nchunks;
                        const size_t cap = W.band[0].cap;                       // a band's share of each list region
                        int32_t* lrgb = rgb_list + (two ? k * cap : 0); int32_t* l13 = list13 + (two ? k * cap : 0); int32_t* l7 = list7 + (two ? k * cap : 0);
                        int32_t* crgb = rgb_count + (k ? 11 : 0); int32_t* c13 = rgb_count + (k ? 12 : 1); int32_t* c7 = rgb_count + (k ? 13 : 2);
                        uint32_t* wins = two ? wins4 + W.band[k].wins : wins4;
                        const int32_t rows = two ? cnt * TPB : (int32_t)n;
                        finish(std::integral_constant<int, 6>{}, nullptr, nullptr, lrgb, crgb, l7, c7, sel, cnt, s);
                        scan_rgb(lrgb, crgb, false, true, 0, ALL, cnt, s);
                        finish(std::integral_constant<int, 5>{}, lrgb, crgb, l13, c13, l7, c7, ALL, cnt, s);
                        scan_rgb(l13, c13, true, false, 1, ALL, cnt, s, wins, rows);
                        finish(std::integral_constant<int, 4>{}, l13, c13, nullptr, nullptr, nullptr, nullptr, ALL, cnt, s, wins, rows);
                        scan_7(l7, c7, 1, cnt, s, wins, rows);
                        finish(std::integral_constant<int, 7>{}, l7, c7, nullptr, nullptr, nullptr, nullptr, ALL, cnt, s, wins, rows);
                    };
                    if (!two) chain(0, st);
                    else {
                        ITW_CHECK(hipEventRecord(aux->fork, st));
                        ITW_CHECK(hipStreamWaitEvent(aux->stream, aux->fork, 0));
                        chain(1, aux->stream);
                        chain(0, st);
                        ITW_CHECK(hipEventRecord(aux->join, aux->stream));
                        ITW_CHECK(hipStreamWaitEvent(st, aux->join, 0));
                    }
                } else {
                    scan_7();
                    finish(std::integral_constant<int, 1>{}, nullptr, nullptr, rgb_list, rgb_count);
                    if (bounded) {
                        scan_rgb(rgb_list, rgb_count, false, true);
                        finish(std::integral_constant<int, 5>{}, rgb_list, rgb_count, list13, count13);
                        scan_rgb(list13, count13, true, false, 1);
                        finish(std::integral_constant<int, 4>{}, list13, count13, nullptr, nullptr);
                    } else {
                        scan_rgb(rgb_list, rgb_count, true, true);
                        finish(std::integral_constant<int, 2>{}, rgb_list, rgb_count, nullptr, nullptr);
                    }
                }
            } else if (bounded && !on7) {
                // RGB profiles whose modes 1/3 scan every shape (`slow`).  Round 4: scan {0,2} -> finish<3> (modes 0,2,4,5,6; lists the blocks
                // whose modes 1/3 an exact bound cannot exclude) -> scan {1,3} over the list -> finish<4>.  Round 5, with a second stream:
                //  * BANDS.  The surface is cut into two interleaved bands (stripes of chunks), one per stream: a band's four dependent
                //    launches leave the chip partly empty at every boundary (a scan wave runs 0.6 ms), and the other band's work fills those
                //    tails (5.23 -> 4.98 ms on the bench surface, 7.30 -> 7.00 on a photograph; four bands are slower again).
                //  * PILOT.  The order pays where few blocks are listed and still costs a little where nearly all are (photographs: 94 %).
                //    Behind band 0's {0,2} scan -- common to both orders -- bc7_pilot_estimate looks at 1/16 of the surface and leaves a
                //    device word; BOTH continuations of each band are enqueued behind it, each gated on that word (ChunkSel.gate): the one
                //    the pilot did not choose returns at once.  No host round trip, no block treated differently from its neighbours.
                //    (A sample encoded ahead on a third stream was measured first: its chain of small launches could not get its share of a
                //    chip the bands' scans fill, and the verdict arrived late: profiles/history/r05/r05d_bc7_pilot_timeline.txt.)
                ITW_CHECK(hipMemsetAsync(rgb_count, 0, 16 * sizeof(int32_t), st));
                const bool two = bc7_bands() > 1 && aux && !aux->single && aux->stream && nchunks >= 32;
                const bool pilot = two && bc7_pilot_threshold() >= 0 && aux->mid;
                int32_t stripe = nchunks / 16;                               // stripes of 64 chunks (16 block rows of a 4096-wide surface), fewer on small surfaces
                stripe = stripe < 1 ? 1 : (stripe > 64 ? 64 : stripe);
                const int32_t stripes = (nchunks + stripe - 1) / stripe;
                struct Band { ChunkSel sel; int32_t cnt; hipStream_t s; int32_t* list; int32_t* count; uint32_t* wins; int32_t rows; uint4* compact; };
                auto band = [&](int k, hipStream_t s) {
                    const ListRegion& r = W.band[k];
                    Band B{ChunkSel{two ? 1 : 0, stripe, k, nullptr, 0}, two ? ((stripes + 1 - k) / 2) * stripe : nchunks, s,
                           reinterpret_cast<int32_t*>(wins4 + r.list), band_count + k, wins4 + r.wins, 0, compact_on ? reinterpret_cast<uint4*>(wins4 + r.compact) : nullptr};
                    B.rows = B.cnt * TPB;                                    // shares x listed blocks <= rows: the list scan's grid covers them (list_scan_parts)
                    if (!two) { B.list = list13; B.wins = wins4; B.rows = (int32_t)n; }     // one band: the global winner rows, both texel regions as one
                    return B;
                };
                auto gated = [&](const Band& B, int32_t want) { ChunkSel g = B.sel; if (pilot) { g.gate = pilot_flag; g.want = want; } return g; };
                auto head = [&](const Band& B) { scan_rgb(nullptr, nullptr, false, true, 0, B.sel, B.cnt, B.s); };
                auto tail = [&](const Band& B) {
                    finish(std::integral_constant<int, 3>{}, nullptr, nullptr, B.list, B.count, nullptr, nullptr, gated(B, 1), B.cnt, B.s, nullptr, 0, nullptr, B.compact);
                    scan_rgb(B.list, B.count, true, false, 1, ALL, B.cnt, B.s, B.wins, B.rows, B.compact);    // an empty list (the other order): returns at once
                    finish(std::integral_constant<int, 4>{}, B.list, B.count, nullptr, nullptr, nullptr, nullptr, ALL, B.cnt, B.s, B.wins, B.rows, B.compact);
                    if (pilot) {
                        scan_rgb(nullptr, nullptr, true, false, 0, gated(B, 0), B.cnt, B.s);
                        finish(std::integral_constant<int, 0>{}, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, gated(B, 0), B.cnt, B.s);
                    }
                };
                if (!two) {
                    const Band B = band(0, st);
                    const bool probe = aux && aux->probe && aux->verdict && aux->verdict->event;
                    if (probe) scan_rgb(nullptr, nullptr, false, true, 0, ChunkSel{2, 1, 0, nullptr, 0}, (nchunks + 7) / 8, st);
                    else head(B);
                    if (aux && aux->verdict && aux->verdict->event) {
                        // a staged run of a host-pointer call (abi.hip): the same estimate, left for the HOST to read under the next
                        // run's upload -- it picks the launch shape of the remaining runs
                        const dim3 grid((unsigned)((B.cnt + 7) / 8));
                        if (L.vec) hipLaunchKernelGGL((bc7_pilot_estimate<true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, wins4, S.skip_mode2 ? 1 : 0, B.sel, pilot_ctr, 256, pilot_flag, aux->verdict->host_counts_dev);
                        else       hipLaunchKernelGGL((bc7_pilot_estimate<false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, wins4, S.skip_mode2 ? 1 : 0, B.sel, pilot_ctr, 256, pilot_flag, aux->verdict->host_counts_dev);
                        ITW_CHECK(hipEventRecord(aux->verdict->event, st));
                        aux->verdict->valid = true;
                    }
                    if (probe) return;
                    tail(B);
                } else {
                    hipStream_t s2 = aux->stream;
                    ITW_CHECK(hipEventRecord(aux->fork, st));                // whatever feeds `src` on st (an upload), and the memset above
                    ITW_CHECK(hipStreamWaitEvent(s2, aux->fork, 0));
                    const Band A = band(0, st), Bb = band(1, s2);
                    head(Bb); head(A);
                    if (pilot) {
                        const dim3 grid((unsigned)((A.cnt + 7) / 8));
                        if (L.vec) hipLaunchKernelGGL((bc7_pilot_estimate<true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, wins4, S.skip_mode2 ? 1 : 0, A.sel, pilot_ctr, bc7_pilot_threshold(), pilot_flag, (int32_t*)nullptr);
                        else       hipLaunchKernelGGL((bc7_pilot_estimate<false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, wins4, S.skip_mode2 ? 1 : 0, A.sel, pilot_ctr, bc7_pilot_threshold(), pilot_flag, (int32_t*)nullptr);
                        ITW_CHECK(hipEventRecord(aux->mid, st));
                        ITW_CHECK(hipStreamWaitEvent(s2, aux->mid, 0));        // band 1's continuations start behind the verdict
                    }
                    tail(Bb); tail(A);
                    ITW_CHECK(hipEventRecord(aux->join, s2));
                    ITW_CHECK(hipStreamWaitEvent(st, aux->join, 0));
                    if (bc7_pilot_debug()) {
                        int32_t h[16];
                        ITW_CHECK(hipStreamSynchronize(st));
                        ITW_CHECK(hipMemcpy(h, rgb_count, sizeof h, hipMemcpyDeviceToHost));
                        std::fprintf(stderr, "bc7 pilot: estimate %d of %d sampled blocks listed -> %s; lists %d + %d of %lld blocks\n", h[8], h[9], h[6] ? "bounded" : "reference order",
                                     h[3], h[4], (long long)n);
                    }
                }
            } else {
                scan_rgb(nullptr, nullptr, true, true);
                scan_7();
                finish(std::integral_constant<int, 0>{}, nullptr, nullptr, nullptr, nullptr);
            }
            return;
        }
    }
    if (S.mode_selection[0]) { launch_search<F_MODES02, 0>(L); launch_finish<F_MODES02>(L); }
    if (S.mode_selection[1] && (S.fastSkipTreshold_mode1 > 0 || S.fastSkipTreshold_mode3 > 0)) {
        if (ranked(S.fastSkipTreshold_mode1) || ranked(S.fastSkipTreshold_mode3)) {
            if (small(S.fastSkipTreshold_mode1) && small(S.fastSkipTreshold_mode3)) launch_search<F_MODES13, 2>(L);
            else launch_search<F_MODES13, 1>(L);
        } else launch_search<F_MODES13, 0>(L);
        launch_finish<F_MODES13>(L);
    }
    if (S.mode_selection[1] && S.fastSkipTreshold_mode7 > 0) {
        if (ranked(S.fastSkipTreshold_mode7)) {
            if (small(S.fastSkipTreshold_mode7)) launch_search<F_MODE7, 2>(L); else launch_search<F_MODE7, 1>(L);
        } else launch_search<F_MODE7, 0>(L);
        launch_finish<F_MODE7>(L);
    }
    if (S.mode_selection[2] || S.mode_selection[3]) launch_finish<F_MODES456>(L);
    if (L.first) ITW_CHECK(hipMemsetAsync(dst, 0, (size_t)n * 16, st));   // no mode enabled: defined (zero) output
}

} // namespace itw
