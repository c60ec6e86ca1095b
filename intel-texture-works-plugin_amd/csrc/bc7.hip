// bc7.hip -- BC7 encoder kernels for gfx950 (MI355X).
//
// Replaces kernel.ispc:616-2037 (CompressBlocksBC7_ispc) behind CompressBlocksBC7
// (ispc_texcomp.cpp:427-430).  Same search as the reference -- PCA line fits per
// subset, p-bit aware endpoint quantisation, two-candidate index selection,
// least-squares refinement of each mode's winner, modes 0-7 -- organised for a
// SIMT machine instead of an SPMD gang:
//
//   * one 4x4 block per lane; the block stays in 16 VGPRs as loaded (packed RGBA8,
//     one v_cvt_f32_ubyteN per use); it is read from HBM with 4 coalesced dwordx4
//     loads per lane and 16 B are written per lane;
//   * one kernel per mode family ({0,2} {1,3} {7} {4,5,6}), run in the reference's
//     order.  The families only communicate through "best error so far"
//     (kernel.ispc:1358, 1638, 1684), which travels in a 4 B/block workspace; a later
//     family overwrites the block only where it wins.  Each family gets its own
//     register allocation and instruction footprint instead of the worst case of all;
//   * shapes are visited in TABLE order wherever the candidate list is the whole
//     table (modes 0/2 always; modes 1/3/7 when their fastSkipTreshold is >= 64, i.e.
//     the `slow` profiles): the shape then lives in SGPRs, subset membership is a
//     scalar branch, and a texel costs work only in the subset it belongs to.  The
//     reference scans its PCA-ranked list with a strict `<`, so among equal errors
//     the lowest rank wins; visiting in table order and breaking error ties by the
//     rank key (part + 64*bound, distinct per shape) selects the same winner;
//   * shorter ranked lists (fast profiles) keep the per-lane order: the i-th entry of
//     the reference's selection sort (kernel.ispc:1365-1384) is the smallest key above
//     the previous one -- a 64-entry LDS scan, no sort, no dynamic register indexing;
//   * fits that do not depend on the mode are shared: shapes 64..79 serve modes 0 and
//     2, every two-subset shape serves modes 1 and 3 (kernel.ispc:1286-1291), and the
//     subset-0 statistics serve both the rank bound and the fit;
//   * during the search only (indices, error, shape, key) of a mode's winner are kept;
//     its endpoint codes are recomputed (deterministically) at commit time.
//
// fp32 VALU bound (GetProfile_slow: ~1e6 separately rounded ops per block against 80
// algorithmic bytes); nothing GEMM shaped, so no MFMA.  Bit-exactness with the oracle
// forbids FMA contraction and any re-association of per-subset float sums.
#include "bcn_core.hpp"
#include "kernels.hpp"

namespace itw {

constexpr int TPB = 64;                   // one wave per workgroup
constexpr float INV255 = 1.0f / 255.0f;   // x/255f under fast-math = x*(1.f/255.f)

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

// ---- endpoint quantisers: quantise, then overwrite the floats with what a decoder reconstructs ----

// modes 0,3,6,7: one p-bit per endpoint, chosen by squared error over `err_ch` channels.  [kernel.ispc:983-1022]
template <int MODE>
__device__ __forceinline__ void quant_pbit(int32_t (&q)[2][4], float (&e)[2][4], int err_ch)
{
    constexpr int BITS = (MODE == 0) ? 4 : (MODE == 7) ? 5 : 7;
    constexpr int L2 = (1 << BITS) * 2 - 1;
    #pragma unroll
    for (int i = 0; i < 2; i++) {
        int32_t qb[2][4];
        float db[2][4];
        #pragma unroll
        for (int b = 0; b < 2; b++)
            #pragma unroll
            for (int p = 0; p < 4; p++) {
                const int32_t v = f2i_x86((e[i][p] * INV255 * (float)L2 - (float)b) * 0.5f + 0.5f) * 2 + b;
                qb[b][p] = iclamp(v, b, L2 - 1 + b);
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
            e[i][p] = (float)((MODE == 0) ? expand_to_byte(q[i][p], 5) : (MODE == 7) ? expand_to_byte(q[i][p], 6) : q[i][p]);
        }
    }
}

// mode 1: one p-bit shared by both endpoints of a subset, RGB error.            [kernel.ispc:1024-1052]
__device__ __forceinline__ void quant_shared_pbit(int32_t (&q)[2][4], float (&e)[2][4])
{
    int32_t qb[2][2][4];
    float db[2][2][4];
    #pragma unroll
    for (int b = 0; b < 2; b++)
        #pragma unroll
        for (int i = 0; i < 2; i++)
            #pragma unroll
            for (int p = 0; p < 4; p++) {
                const int32_t v = f2i_x86((e[i][p] * INV255 * 127.0f - (float)b) * 0.5f + 0.5f) * 2 + b;
                qb[b][i][p] = iclamp(v, b, 126 + b);
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
            e[i][p] = (float)expand_to_byte(q[i][p], 7);
        }
}

// modes 2,4 (5 bits) and 5 (7 bits): plain rounding.                            [kernel.ispc:1054-1065]
template <int BITS>
__device__ __forceinline__ void quant_plain(int32_t (&q)[2][4], float (&e)[2][4])
{
    constexpr int L = 1 << BITS;
    #pragma unroll
    for (int i = 0; i < 2; i++)
        #pragma unroll
        for (int p = 0; p < 4; p++) {
            q[i][p] = iclamp(f2i_x86(e[i][p] * INV255 * (float)(L - 1) + 0.5f), 0, L - 1);
            e[i][p] = (float)expand_to_byte(q[i][p], BITS);
        }
}

template <int MODE>
__device__ __forceinline__ void quant_mode(int32_t (&q)[2][4], float (&e)[2][4], int err_ch)
{
    if (MODE == 0 || MODE == 3 || MODE == 6 || MODE == 7) quant_pbit<MODE>(q, e, err_ch);
    else if (MODE == 1) quant_shared_pbit(q, e);
    else if (MODE == 2 || MODE == 4) quant_plain<5>(q, e);
    else quant_plain<7>(q, e);
}

// ---- per-lane state ---------------------------------------------------------------------------
struct Lane {
    TexU8 tex;
    float best_err;
    float opaque_err;
    uint32_t best[4];
    bool improved;
    SeedTables T;
    int32_t* keys;            // LDS column (ranked-list path only): keys[i * TPB]
};

struct Win {                  // winner of one multi-subset mode during the search
    uint32_t qb[2];
    float err;
    int32_t shape;            // table index 0..63 (two subsets) / 64..127 (three)
    int32_t key;              // rank key of the shape (tie-break in table-order scans)
};

__device__ __forceinline__ void reset(Win& w, int shape0)
{
    w.qb[0] = w.qb[1] = 0u;
    w.err = __builtin_inff();
    w.shape = shape0;
    w.key = 0x7fffffff;
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

// Quantise a fitted shape for MODE and pick indices.  Keeps the candidate if its error is lower, or equal
// with a lower rank key: in a table-order scan that is the candidate the reference's ranked, strict-`<`
// scan keeps; in a ranked scan keys increase, so the second clause never fires.   [kernel.ispc:1279-1327]
template <int MODE>
__device__ __forceinline__ void try_shape(Win& best, const Lane& ln, const float (&fit)[3][2][4], const Shape& sh, int shape_index, int32_t key)
{
    constexpr ModeTraits M = traits(MODE);
    float ep[3][2][4];
    #pragma unroll
    for (int j = 0; j < M.pairs; j++) {
        int32_t q[2][4];
        #pragma unroll
        for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) ep[j][i][p] = fit[j][i][p];
        quant_mode<MODE>(q, ep[j], M.ch);
    }
    uint32_t qb[2];
    const float err = select_indices<M.bits, M.ch, false>(qb, ln.tex, ep, sh.pattern);
    if (err < best.err || (err == best.err && key < best.key)) {
        best.qb[0] = qb[0]; best.qb[1] = qb[1];
        best.err = err;
        best.shape = shape_index;
        best.key = key;
    }
}

// Least-squares refinement of a mode's winner, then the mode competes for the block.  [kernel.ispc:1329-1362]
template <int MODE>
__device__ __forceinline__ void refine_and_commit(Lane& ln, Win& w, int iterations, int settings_channels)
{
    constexpr ModeTraits M = traits(MODE);
    const Shape sh = load_shape(w.shape);
    // endpoint codes of the search-time winner: refit + quantise its shape again (same inputs, same bits)
    int32_t cq[3][2][4];
    {
        float ep[3][2][4];
        #pragma unroll
        for (int j = 0; j < M.pairs; j++) {
            ep[j][0][3] = 0.f; ep[j][1][3] = 0.f;
            fit_subset<M.ch, true>(ep[j], ln.tex, subset_mask(sh, j), ln.T);
            quant_mode<MODE>(cq[j], ep[j], M.ch);
        }
        if (M.pairs == 2) for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) cq[2][i][p] = 0;
    }
    for (int it = 0; it < iterations; it++) {
        ln.tex.fence();
        float ep[3][2][4];
        int32_t q[3][2][4];
        #pragma unroll
        for (int j = 0; j < M.pairs; j++) {
            ep[j][0][3] = 0.f; ep[j][1][3] = 0.f;
            refit_subset<M.bits, M.ch>(ep[j], ln.tex, w.qb, subset_mask(sh, j), ln.T);
            quant_mode<MODE>(q[j], ep[j], settings_channels);          // :1343 passes the profile's channel count
        }
        uint32_t qb[2];
        const float err = select_indices<M.bits, M.ch, false>(qb, ln.tex, ep, sh.pattern);
        if (err < w.err) {
            #pragma unroll
            for (int j = 0; j < M.pairs; j++) for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) cq[j][i][p] = q[j][i][p];
            w.qb[0] = qb[0]; w.qb[1] = qb[1];
            w.err = err;
        }
    }
    float err = w.err;
    if (MODE != 7) err += ln.opaque_err;
    if (err < ln.best_err) {
        ln.best_err = err;
        ln.improved = true;
        emit_multi<MODE>(ln.best, cq, w.qb, w.shape);
    }
}

// modes 0 and 2: three subsets; shapes in table order (wave-uniform), one fit per shape.  [kernel.ispc:1386-1394]
__device__ __forceinline__ void modes_02(Lane& ln, const bc7_enc_settings& S)
{
    Win b0, b2;
    reset(b0, 64); reset(b2, 64);
    const int count = S.skip_mode2 ? 16 : 64;
    for (int part = 0; part < count; part++) {
        ln.tex.fence();
        const Shape sh = load_shape(64 + part);
        float fit[3][2][4];
        #pragma unroll
        for (int j = 0; j < 3; j++) {
            fit_subset<3, true>(fit[j], ln.tex, subset_mask(sh, j), ln.T);
            fit[j][0][3] = 0.f; fit[j][1][3] = 0.f;      // the reference's unwritten alpha slots, pinned to 0
        }
        // list order == table order here: key = position
        if (part < 16) try_shape<0>(b0, ln, fit, sh, 64 + part, part);
        if (!S.skip_mode2) try_shape<2>(b2, ln, fit, sh, 64 + part, part);
    }
    refine_and_commit<0>(ln, b0, S.refineIterations[0], S.channels);
    if (!S.skip_mode2) refine_and_commit<2>(ln, b2, S.refineIterations[2], S.channels);
}

// Two-subset modes.  FAMILY7 = false: modes 1 and 3 (3-channel fit, shared);  true: mode 7 (4-channel fit).
// RANK_CH: channels used by the PCA ranking (3 for modes 1/3; the profile's channel count for mode 7).
template <bool FAMILY7, int RANK_CH>
__device__ __forceinline__ void two_subset_modes(Lane& ln, const bc7_enc_settings& S)
{
    constexpr int FIT_CH = FAMILY7 ? 4 : 3;
    const int na = FAMILY7 ? S.fastSkipTreshold_mode7 : S.fastSkipTreshold_mode1;   // first mode of the family
    const int nb = FAMILY7 ? 0 : S.fastSkipTreshold_mode3;                           // second mode
    if (na <= 0 && nb <= 0) return;
    Win wa, wb;
    reset(wa, 0); reset(wb, 0);

    Stats<RANK_CH> full;
    stats_of<RANK_CH>(full, ln.tex, 0xffffu);

    const bool whole_table = (na <= 0 || na >= 64) && (nb <= 0 || nb >= 64);
    if (whole_table) {
        // every shape is a candidate: table order, rank key only breaks ties
        for (int part = 0; part < 64; part++) {
            ln.tex.fence();
            const Shape sh = load_shape(part);
            const uint32_t m0 = sh.masks & 0xffffu, m1 = sh.masks >> 16;
            float fit[3][2][4];
            int32_t key;
            if constexpr (RANK_CH == FIT_CH) {
                Stats<FIT_CH> s0;
                stats_of<FIT_CH>(s0, ln.tex, m0);
                key = (int32_t)((uint32_t)part + (uint32_t)split_bound_from<FIT_CH>(s0, reinterpret_cast<const Stats<FIT_CH>&>(full), ln.T) * 64u);
                fit_from_stats<FIT_CH, true>(fit[0], ln.tex, m0, s0, ln.T);
            } else {
                key = (int32_t)((uint32_t)part + (uint32_t)split_bound<RANK_CH>(ln.tex, m0, full, ln.T) * 64u);
                fit_subset<FIT_CH, true>(fit[0], ln.tex, m0, ln.T);
            }
            fit_subset<FIT_CH, true>(fit[1], ln.tex, m1, ln.T);
            if (FIT_CH == 3) for (int j = 0; j < 2; j++) { fit[j][0][3] = 0.f; fit[j][1][3] = 0.f; }
            if (FAMILY7) {
                try_shape<7>(wa, ln, fit, sh, part, key);
            } else {
                if (na > 0) try_shape<1>(wa, ln, fit, sh, part, key);
                if (nb > 0) try_shape<3>(wb, ln, fit, sh, part, key);
            }
        }
    } else {
        // ranked prefix: keys to LDS, then walk them in increasing order per lane        [kernel.ispc:1400-1414]
        for (int part = 0; part < 64; part++) {
            ln.tex.fence();
            const uint32_t m0 = BCN_SUBSET_MASKS[part] & 0xffffu;
            const int32_t bound = split_bound<RANK_CH>(ln.tex, m0, full, ln.T);
            ln.keys[part * TPB] = (int32_t)((uint32_t)part + (uint32_t)bound * 64u);
        }
        const int n = min(max(na, nb), 64);
        int32_t prev = 0;
        for (int i = 0; i < n; i++) {
            int32_t cur = 0x7fffffff;
            for (int t = 0; t < 64; t++) {
                const int32_t k = ln.keys[t * TPB];
                if ((i == 0 || k > prev) && k <= cur) cur = k;
            }
            prev = cur;
            ln.tex.fence();
            const int shape = prev & 63;
            const Shape sh = load_shape(shape);
            float fit[3][2][4];
            #pragma unroll
            for (int j = 0; j < 2; j++) {
                fit_subset<FIT_CH, true>(fit[j], ln.tex, subset_mask(sh, j), ln.T);
                if (FIT_CH == 3) { fit[j][0][3] = 0.f; fit[j][1][3] = 0.f; }
            }
            if (FAMILY7) {
                try_shape<7>(wa, ln, fit, sh, shape, prev);
            } else {
                if (i < na) try_shape<1>(wa, ln, fit, sh, shape, prev);
                if (i < nb) try_shape<3>(wb, ln, fit, sh, shape, prev);
            }
        }
    }
    if (FAMILY7) {
        refine_and_commit<7>(ln, wa, S.refineIterations[7], S.channels);
    } else {
        if (na > 0) refine_and_commit<1>(ln, wa, S.refineIterations[1], S.channels);
        if (nb > 0) refine_and_commit<3>(ln, wb, S.refineIterations[3], S.channels);
    }
}

// ---- modes 4 and 5: vector part (3 channels) + one separately coded channel ------------------

// scalar channel: min/max endpoints, then `iters` rounds of LS refit.             [kernel.ispc:1437-1563]
template <int BITS, int EPBITS>
__device__ __forceinline__ float encode_scalar(uint32_t (&qb)[2], int32_t (&qe)[2], const float (&v)[16], int iters, const SeedTables& T)
{
    constexpr int LEVELS = 1 << BITS;
    constexpr float L1 = (float)(LEVELS - 1);
    constexpr int EL = 1 << EPBITS;
    float ep[2] = {255.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k++) { ep[0] = fmin_x86(ep[0], v[k]); ep[1] = fmax_x86(ep[1], v[k]); }
    float err = 0.f;
    for (int round = 0; ; round++) {
        #pragma unroll
        for (int i = 0; i < 2; i++) {                                            // channel_quant_dequant
            qe[i] = iclamp(f2i_x86(ep[i] * INV255 * (float)(EL - 1) + 0.5f), 0, EL - 1);
            ep[i] = (float)expand_to_byte(qe[i], EPBITS);
        }
        qb[0] = qb[1] = 0u;                                                      // channel_opt_quant
        err = 0.f;
        const float rspan = ispc_rcp(ep[1] - ep[0] + 0.001f, T);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float proj = (v[k] - ep[0]) * rspan;
            int32_t q1 = iclamp(f2i_x86(proj * (float)LEVELS + 0.5f), 1, LEVELS - 1);
            const float w0 = (float)weight_of<BITS>(q1 - 1), w1 = (float)weight_of<BITS>(q1);
            const float d0 = (float)f2i_x86(((64.0f - w0) * ep[0] + w0 * ep[1] + 32.0f) * 0.015625f);
            const float d1 = (float)f2i_x86(((64.0f - w1) * ep[0] + w1 * ep[1] + 32.0f) * 0.015625f);
            float e0 = 0.f, e1 = 0.f;
            e0 += sq(d0 - v[k]);
            e1 += sq(d1 - v[k]);
            const bool first = e0 < e1;
            const uint32_t q = (uint32_t)(first ? q1 - 1 : q1);
            if (k < 8) qb[0] += q << (4 * k); else qb[1] += q << (4 * (k - 8));
            err += (float)(int32_t)(first ? e0 : e1);
        }
        if (round >= iters) break;
        float atb1 = 0.f, sum_q = 0.f, sum_qq = 0.f, sum = 0.f;                   // channel_opt_endpoints
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float q = (float)((k < 8 ? qb[0] >> (4 * k) : qb[1] >> (4 * (k - 8))) & 15u);
            const float x = L1 - q;
            sum_q += q; sum_qq += q * q;
            sum += v[k];
            atb1 += x * v[k];
        }
        const float atb2 = L1 * sum - atb1;
        const float cxx = 16.0f * (L1 * L1) - (2.0f * L1) * sum_q + sum_qq;
        const float cyy = sum_qq;
        const float cxy = L1 * sum_q - sum_qq;
        const float det = cxx * cyy - cxy * cxy;
        const float scale = L1 * ispc_rcp(det, T);
        ep[0] = fclamp_x86((atb1 * cyy - atb2 * cxy) * scale, 0.f, 255.f);
        ep[1] = fclamp_x86((atb2 * cxx - atb1 * cxy) * scale, 0.f, 255.f);
        if (fabsf(det) < 0.001f) { ep[0] = sum * 0.0625f; ep[1] = ep[0]; }
    }
    return err;
}

// one (mode, rotation, index-swap) candidate                                       [kernel.ispc:1565-1621]
template <int MODE, int SWAP>
__device__ __forceinline__ void try_dual(Dual& best, float& best_err, const Lane& ln, const bc7_enc_settings& S, int rotation)
{
    constexpr int BITS = SWAP ? 3 : 2;
    constexpr int ABITS = SWAP ? 2 : ((MODE == 4) ? 3 : 2);
    constexpr int AEPB = (MODE == 4) ? 6 : 8;

    // rotated colour block: channel `rotation` is replaced by alpha (RGBA profile) or 255 (RGB profile);
    // the displaced channel is coded separately
    TexU8 rot;
    float scalar[16];
    const uint32_t sh8 = 8u * (uint32_t)rotation;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t w = ln.tex.w[k];
        scalar[k] = (float)((w >> sh8) & 255u);
        uint32_t r = w;
        if (rotation < 3) {
            const uint32_t fill = (S.channels == 4) ? (w >> 24) : 255u;
            r = (w & ~(255u << sh8)) | (fill << sh8);
        }
        rot.w[k] = r;
    }

    float ep[3][2][4];
    int32_t q[2][4];
    uint32_t qb[2];
    ep[0][0][3] = 0.f; ep[0][1][3] = 0.f;
    fit_subset<3, true>(ep[0], rot, 0xffffu, ln.T);
    quant_mode<MODE>(q, ep[0], 3);
    float err = select_indices<BITS, 3, false>(qb, rot, ep, 0u);
    const int iters = S.refineIterations[MODE];
    for (int it = 0; it < iters; it++) {
        refit_subset<BITS, 3>(ep[0], rot, qb, 0xffffu, ln.T);
        quant_mode<MODE>(q, ep[0], 3);
        err = select_indices<BITS, 3, false>(qb, rot, ep, 0u);
    }

    int32_t aq[2];
    uint32_t aqb[2];
    err += encode_scalar<ABITS, AEPB>(aqb, aq, scalar, S.refineIterations_channel, ln.T);

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

__device__ __forceinline__ void modes_45(Lane& ln, const bc7_enc_settings& S)      // [kernel.ispc:1623-1655]
{
    Dual best;
    #pragma unroll
    for (int i = 0; i < 2; i++) for (int p = 0; p < 4; p++) best.q[i][p] = 0;
    best.qb[0] = best.qb[1] = best.aqb[0] = best.aqb[1] = 0u;
    best.aq[0] = best.aq[1] = 0; best.rotation = 0; best.swap = 0;
    float best_err = ln.best_err;

    for (int r = S.mode45_channel0; r < S.channels; r++) {
        try_dual<4, 0>(best, best_err, ln, S, r);
        try_dual<4, 1>(best, best_err, ln, S, r);
    }
    if (best_err < ln.best_err) { ln.best_err = best_err; ln.improved = true; emit_dual<4>(ln.best, best); }

    for (int r = S.mode45_channel0; r < S.channels; r++)
        try_dual<5, 0>(best, best_err, ln, S, r);
    if (best_err < ln.best_err) { ln.best_err = best_err; ln.improved = true; emit_dual<5>(ln.best, best); }
}

// ---- mode 6: one subset, RGBA, 4-bit indices                                     [kernel.ispc:1657-1689]
template <int CH>
__device__ __forceinline__ void mode_6(Lane& ln, const bc7_enc_settings& S)
{
    float ep[3][2][4];
    int32_t q[2][4];
    uint32_t qb[2];
    ep[0][0][3] = 0.f; ep[0][1][3] = 0.f;
    fit_subset<CH, true>(ep[0], ln.tex, 0xffffu, ln.T);
    if (CH == 3) { ep[0][0][3] = 255.f; ep[0][1][3] = 255.f; }
    quant_mode<6>(q, ep[0], CH);
    float err = select_indices<4, CH, false>(qb, ln.tex, ep, 0u);
    const int iters = S.refineIterations[6];
    for (int it = 0; it < iters; it++) {
        refit_subset<4, CH>(ep[0], ln.tex, qb, 0xffffu, ln.T);
        quant_mode<6>(q, ep[0], CH);
        err = select_indices<4, CH, false>(qb, ln.tex, ep, 0u);
    }
    if (err < ln.best_err) {
        ln.best_err = err;
        ln.improved = true;
        emit_mode6(ln.best, q, qb);
    }
}

// ---- kernels: one per mode family ------------------------------------------------------------------
enum Family { F_MODES02 = 0, F_MODES13 = 1, F_MODE7 = 2, F_MODES456 = 3 };

template <int FAMILY, bool VEC16>
__global__ void __launch_bounds__(TPB)
bc7_family_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks,
                  uint8_t* __restrict__ dst, float* __restrict__ err_ws, const bc7_enc_settings S, const int first)
{
    __shared__ int32_t s_keys[(FAMILY == F_MODES13 || FAMILY == F_MODE7) ? 64 * TPB : 1];
    const int32_t b = blockIdx.x * TPB + threadIdx.x;
    if (b >= nblocks) return;
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;

    Lane ln;
    ln.T = global_seed_tables();
    ln.keys = s_keys + ((FAMILY == F_MODES13 || FAMILY == F_MODE7) ? threadIdx.x : 0);
    const uint8_t* p = src + (int64_t)yy * 4 * stride + (int64_t)xx * 16;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        if (VEC16) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + y * stride);
            ln.tex.w[y * 4 + 0] = v.x; ln.tex.w[y * 4 + 1] = v.y; ln.tex.w[y * 4 + 2] = v.z; ln.tex.w[y * 4 + 3] = v.w;
        } else {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p + y * stride);
            #pragma unroll
            for (int x = 0; x < 4; x++) ln.tex.w[y * 4 + x] = q[x];
        }
    }

    ln.best_err = first ? __builtin_inff() : err_ws[b];
    ln.best[0] = ln.best[1] = ln.best[2] = ln.best[3] = 0u;
    ln.improved = (first != 0);                    // the first family always defines the block
    ln.opaque_err = 0.f;                                                           // kernel.ispc:1267-1277
    if (S.channels == 4) {
        float e = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) e += sq(ln.tex.get(3, k) - 255.0f);
        ln.opaque_err = e;
    }

    if (FAMILY == F_MODES02) modes_02(ln, S);
    if (FAMILY == F_MODES13) two_subset_modes<false, 3>(ln, S);
    if (FAMILY == F_MODE7) { if (S.channels == 4) two_subset_modes<true, 4>(ln, S); else two_subset_modes<true, 3>(ln, S); }
    if (FAMILY == F_MODES456) {
        if (S.mode_selection[2]) modes_45(ln, S);
        if (S.mode_selection[3]) { if (S.channels == 4) mode_6<4>(ln, S); else mode_6<3>(ln, S); }
    }

    if (ln.improved) {
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (int64_t)b * 16);
        if (VEC16) *reinterpret_cast<uint4*>(d) = make_uint4(ln.best[0], ln.best[1], ln.best[2], ln.best[3]);
        else { d[0] = ln.best[0]; d[1] = ln.best[1]; d[2] = ln.best[2]; d[3] = ln.best[3]; }
        err_ws[b] = ln.best_err;
    }
}

template <int FAMILY>
static void launch_family(bool vec, dim3 grid, hipStream_t st, const uint8_t* src, int64_t stride, int bx, int32_t n,
                          uint8_t* dst, float* ws, const bc7_enc_settings& S, int first)
{
    if (vec) hipLaunchKernelGGL((bc7_family_kernel<FAMILY, true>),  grid, dim3(TPB), 0, st, src, stride, bx, n, dst, ws, S, first);
    else     hipLaunchKernelGGL((bc7_family_kernel<FAMILY, false>), grid, dim3(TPB), 0, st, src, stride, bx, n, dst, ws, S, first);
}

size_t bc7_workspace_bytes(int width, int height)
{
    return (size_t)(width / 4) * (size_t)(height / 4) * sizeof(float);
}

// Families run in the reference's order (kernel.ispc:1970-1977): {0,2} -> {1,3} -> {7} -> {4,5} -> {6}.
void launch_bc7(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst,
                const bc7_enc_settings& s, float* err_ws, hipStream_t st)
{
    const int bx = width / 4, by = height / 4;
    const int64_t n = (int64_t)bx * by;
    if (n <= 0) return;
    bc7_enc_settings S = s;
    S.channels = (s.channels == 4) ? 4 : 3;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const dim3 grid((unsigned)((n + TPB - 1) / TPB));
    int first = 1;
    if (S.mode_selection[0]) { launch_family<F_MODES02>(vec, grid, st, src, stride, bx, (int32_t)n, dst, err_ws, S, first); first = 0; }
    if (S.mode_selection[1] && (S.fastSkipTreshold_mode1 > 0 || S.fastSkipTreshold_mode3 > 0)) {
        launch_family<F_MODES13>(vec, grid, st, src, stride, bx, (int32_t)n, dst, err_ws, S, first); first = 0;
    }
    if (S.mode_selection[1] && S.fastSkipTreshold_mode7 > 0) {
        launch_family<F_MODE7>(vec, grid, st, src, stride, bx, (int32_t)n, dst, err_ws, S, first); first = 0;
    }
    if (S.mode_selection[2] || S.mode_selection[3]) {
        launch_family<F_MODES456>(vec, grid, st, src, stride, bx, (int32_t)n, dst, err_ws, S, first); first = 0;
    }
    if (first) (void)hipMemsetAsync(dst, 0, (size_t)n * 16, st);   // no mode enabled: defined (zero) output
}

} // namespace itw
