// bcn_core.hpp -- device building blocks shared by the BC7 and BC6H kernels (gfx950).
//
// What the reference calls block_segment / block_pca_bound_split / block_quant /
// opt_endpoints (kernel.ispc:763-971, 1133-1262) lives here, re-expressed for a
// SIMT lane:
//   * texels of the lane's block sit in registers as px[channel][texel];
//   * a subset is a 16-bit texel mask; loops are written `if (mask bit k) {...}`.
//     When the mask is wave-uniform (partition chosen by a loop counter) hipcc
//     turns that into scalar branches and the masked-out texels cost nothing;
//     when it differs per lane (ranked partition lists, refinement of the lane's
//     own winner) it becomes exec masking.  Skipping a texel is bit-identical to
//     the reference's multiply-by-0/1-flag formulation: all texel values are
//     non-negative finite, every accumulator starts at +0 and x + (+0) == x.
//   * every float sum runs in texel order inside the lane (no cross-lane float
//     reduction anywhere), every divide is lowered as catalogued in SURVEY 8c.
#pragma once
#include "x86_math.hpp"

namespace itw {

#define BCN_TABLE_QUAL __device__ const
#include "bc7_tables.h"
#undef BCN_TABLE_QUAL

// v_min_f32 / v_max_f32 as instructions (operands known to be numbers: no NaN quieting needed)
__device__ __forceinline__ float vmin_raw(float a, float b) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }
__device__ __forceinline__ float vmax_raw(float a, float b) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); return d; }

// ---- partition table access (kernel.ispc:688-758) --------------------------
struct Shape {
    uint32_t pattern;   // 2 bits / texel
    uint32_t masks;     // subset0 | subset1 << 16
    uint32_t anchors;   // anchor(subset1) << 4 | anchor(subset2)
};

__device__ __forceinline__ Shape load_shape(int table_index)
{
    return Shape{BCN_PATTERN[table_index], BCN_SUBSET_MASKS[table_index], (uint32_t)BCN_ANCHORS[table_index]};
}

__device__ __forceinline__ uint32_t subset_mask(const Shape& s, int j)
{
    const uint32_t m0 = s.masks & 0xffffu, m1 = s.masks >> 16;
    return (j == 0) ? m0 : ((j == 1) ? m1 : (~m0 & ~m1 & 0xffffu));
}

// ---- texel storage ------------------------------------------------------------
// TexF : planar floats (BC6H: uf16-domain values).  TexU8: the block as loaded, 16 packed RGBA8 words; a channel
// value is one v_cvt_f32_ubyteN at the point of use -- 16 VGPRs instead of 64 for the whole search.
struct TexF {
    float v[4][16];
    __device__ __forceinline__ float get(int p, int k) const { return v[p][k]; }
    __device__ __forceinline__ void fence()
    {
#pragma unroll
        for (int p = 0; p < 3; p++)
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("" : "+v"(v[p][k]));
    }
};
struct TexU8 {
    uint32_t w[16];
    __device__ __forceinline__ float get(int p, int k) const { return (float)((w[k] >> (8 * p)) & 255u); }
    // Compiler fence for candidate loops: everything derived from the texels (conversions, products r*r, r*g ...)
    // is loop invariant, and LICM would hoist ~200 such values out of the shape loop and keep them in registers
    // (or spill them).  Declaring the words "modified" at the top of an iteration keeps the working set at the
    // 16 packed words; no instruction is emitted.
    __device__ __forceinline__ void fence()
    {
#pragma unroll
        for (int k = 0; k < 16; k++) asm volatile("" : "+v"(w[k]));
    }
};

// ---- second-moment statistics of a subset (kernel.ispc:763-803) -------------
template <int CH>
struct Stats {
    float m[10];   // packed upper triangle of the 4x4 moment matrix: 00 01 02 03 11 12 13 22 23 33
    float s[4];    // channel sums
    float n;       // texel count
};

template <int CH, class TX>
__device__ __forceinline__ void stats_of(Stats<CH>& st, const TX& px, uint32_t mask)
{
    #pragma unroll
    for (int i = 0; i < 10; i++) st.m[i] = 0.f;
    #pragma unroll
    for (int i = 0; i < 4; i++) st.s[i] = 0.f;
    st.n = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((mask >> k) & 1u) {
            const float r = px.get(0, k), g = px.get(1, k), b = px.get(2, k);
            st.n += 1.0f;
            st.s[0] += r; st.s[1] += g; st.s[2] += b;
            st.m[0] += r * r; st.m[1] += r * g; st.m[2] += r * b;
            st.m[4] += g * g; st.m[5] += g * b;
            st.m[7] += b * b;
            if (CH == 4) {
                const float a = px.get(3, k);
                st.s[3] += a;
                st.m[3] += r * a; st.m[6] += g * a; st.m[8] += b * a; st.m[9] += a * a;
            }
        }
    }
}

// covariance = moments - sum*sum*rcp(n)                       (kernel.ispc:805-823)
template <int CH>
__device__ __forceinline__ void covariance_of(float (&cv)[10], const Stats<CH>& st, float rn)
{
    cv[0] = st.m[0] - st.s[0] * st.s[0] * rn;
    cv[1] = st.m[1] - st.s[0] * st.s[1] * rn;
    cv[2] = st.m[2] - st.s[0] * st.s[2] * rn;
    cv[4] = st.m[4] - st.s[1] * st.s[1] * rn;
    cv[5] = st.m[5] - st.s[1] * st.s[2] * rn;
    cv[7] = st.m[7] - st.s[2] * st.s[2] * rn;
    if (CH == 4) {
        cv[3] = st.m[3] - st.s[0] * st.s[3] * rn;
        cv[6] = st.m[6] - st.s[1] * st.s[3] * rn;
        cv[8] = st.m[8] - st.s[2] * st.s[3] * rn;
        cv[9] = st.m[9] - st.s[3] * st.s[3] * rn;
    } else {
        cv[3] = cv[6] = cv[8] = cv[9] = 0.f;
    }
}

// Power iteration from (1,..,1), renormalised (rsqrt) after every second step.  (kernel.ispc:207-229)
template <int CH, int ITERS, bool FAST = false>
__device__ __forceinline__ void principal_axis(float (&v)[4], const float (&cv)[10], const SeedTables& T)
{
    v[0] = v[1] = v[2] = v[3] = 1.f;
#pragma unroll
    for (int it = 0; it < ITERS; it++) {
        float a[4];
        if (CH == 3) {
            a[0] = cv[0] * v[0] + cv[1] * v[1] + cv[2] * v[2];
            a[1] = cv[1] * v[0] + cv[4] * v[1] + cv[5] * v[2];
            a[2] = cv[2] * v[0] + cv[5] * v[1] + cv[7] * v[2];
            a[3] = 0.f;
        } else {
            a[0] = cv[0] * v[0] + cv[1] * v[1] + cv[2] * v[2] + cv[3] * v[3];
            a[1] = cv[1] * v[0] + cv[4] * v[1] + cv[5] * v[2] + cv[6] * v[3];
            a[2] = cv[2] * v[0] + cv[5] * v[1] + cv[7] * v[2] + cv[8] * v[3];
            a[3] = cv[3] * v[0] + cv[6] * v[1] + cv[8] * v[2] + cv[9] * v[3];
        }
        #pragma unroll
        for (int p = 0; p < CH; p++) v[p] = a[p];
        if (it & 1) {
            float nsq = a[0] * a[0];                      // the reference's 0 + a0*a0: a square is never -0, so the sum is the square
            #pragma unroll
            for (int p = 1; p < CH; p++) nsq += a[p] * a[p];
            const float rn = ispc_rsqrt<FAST>(nsq, T);
            #pragma unroll
            for (int p = 0; p < CH; p++) v[p] *= rn;
        }
    }
}

// PCA line fit of one subset: endpoints = mean + extreme projections * axis.
// CLAMP255: BC7 clamps to [0,255] (kernel.ispc:896-905); BC6H keeps the raw fit (857-894).
// ep[0][p] / ep[1][p]: low / high endpoint.  Slots p >= CH are left untouched.
template <int CH, bool CLAMP255, bool FAST = false, class TX>
__device__ __forceinline__ void fit_from_stats(float (&ep)[2][4], const TX& px, uint32_t mask, const Stats<CH>& st, const SeedTables& T)
{
    const float rn = ispc_rcp(st.n, T);
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
    principal_axis<CH, 8, FAST>(axis, cv, T);

    // Extreme projections.  Two exact simplifications, as in bc7_exact.hpp fit_line: (1) the reference's `dot = 0; dot += ..`
    // starts from its first product -- 0 + x differs from x only in the sign of a zero, and a zero of either sign in lo / hi
    // gives the same endpoints ((+-0) * axis + dc with dc >= +0: texels here are non-negative finite numbers); (2) minps /
    // maxps equal v_min / v_max_f32 unless a NaN is involved (again up to the sign of a zero), texels and dc are finite,
    // so only lanes whose axis is not finite (degenerate normalisation) take the compare-and-select form.
    float lo = __builtin_inff(), hi = -__builtin_inff();
    float probe = axis[0];
    #pragma unroll
    for (int p = 1; p < CH; p++) probe += axis[p];
    probe *= 0.0f;                                                              // NaN iff some axis component is NaN or inf
    if (__builtin_expect(probe != probe, 0)) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if ((mask >> k) & 1u) {
                float dot = axis[0] * (px.get(0, k) - dc[0]);
                #pragma unroll
                for (int p = 1; p < CH; p++) dot += axis[p] * (px.get(p, k) - dc[p]);
                lo = fmin_x86(lo, dot);
                hi = fmax_x86(hi, dot);
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if ((mask >> k) & 1u) {
                float dot = axis[0] * (px.get(0, k) - dc[0]);
                #pragma unroll
                for (int p = 1; p < CH; p++) dot += axis[p] * (px.get(p, k) - dc[p]);
                lo = vmin_raw(lo, dot);                     // (fminf() would add a canonicalising v_max(x, x) per loop-carried operand)
                hi = vmax_raw(hi, dot);
            }
        }
    }
    if (hi - lo < 1.0f) { lo -= 0.5f; hi += 0.5f; }
    #pragma unroll
    for (int p = 0; p < CH; p++) {
        float a = lo * axis[p] + dc[p], b = hi * axis[p] + dc[p];
        if (CLAMP255) { a = fclamp_x86(a, 0.f, 255.f); b = fclamp_x86(b, 0.f, 255.f); }
        ep[0][p] = a; ep[1][p] = b;
    }
}

template <int CH, bool CLAMP255, bool FAST = false, class TX>
__device__ __forceinline__ void fit_subset(float (&ep)[2][4], const TX& px, uint32_t mask, const SeedTables& T)
{
    Stats<CH> st;
    stats_of<CH>(st, px, mask);
    fit_from_stats<CH, CLAMP255, FAST>(ep, px, mask, st, T);
}

// trace - largest eigenvalue (4 power iterations) of a scaled covariance.   (kernel.ispc:907-939)
template <int CH, bool FAST = false>
__device__ __forceinline__ float pca_residual(float (&cv)[10], const SeedTables& T)
{
    const float inv_var = 1.0f / 65536.0f;
    #pragma unroll
    for (int i = 0; i < 10; i++) cv[i] *= inv_var;
    const float eps = 0.001f * 0.001f;
    cv[0] += eps; cv[4] += eps; cv[7] += eps;           // not cv[9]: reference quirk
    float axis[4];
    principal_axis<CH, 4, FAST>(axis, cv, T);
    float w[4];
    if (CH == 3) {
        w[0] = cv[0] * axis[0] + cv[1] * axis[1] + cv[2] * axis[2];
        w[1] = cv[1] * axis[0] + cv[4] * axis[1] + cv[5] * axis[2];
        w[2] = cv[2] * axis[0] + cv[5] * axis[1] + cv[7] * axis[2];
    } else {
        w[0] = cv[0] * axis[0] + cv[1] * axis[1] + cv[2] * axis[2] + cv[3] * axis[3];
        w[1] = cv[1] * axis[0] + cv[4] * axis[1] + cv[5] * axis[2] + cv[6] * axis[3];
        w[2] = cv[2] * axis[0] + cv[5] * axis[1] + cv[7] * axis[2] + cv[8] * axis[3];
        w[3] = cv[3] * axis[0] + cv[6] * axis[1] + cv[8] * axis[2] + cv[9] * axis[3];
    }
    float sq_sum = sq(w[0]);                              // 0 + w0*w0 of the reference: a square is never -0
    #pragma unroll
    for (int p = 1; p < CH; p++) sq_sum += sq(w[p]);
    const float lambda = sqrtf(sq_sum);
    float bound = cv[0] + cv[4] + cv[7];
    if (CH == 4) bound += cv[9];
    bound -= lambda;
    return fmax_x86(bound, 0.0f);
}

// Lower bound on the two-subset error of a shape, as an integer sort key component:
// (int)(sqrt(res(subset0) + res(rest)) * 256), rest = full - subset0.   (kernel.ispc:952-971, 1404-1409)
// rn_a / rn_b: ISPC rcp of the two texel counts when the caller already has them (wave-uniform shapes read them from
// the 17-entry table of bc7_exact.hpp instead of running the seed + Newton emulation twice per shape); < 0 = compute here.
template <int CH, bool FAST = false>
__device__ __forceinline__ int32_t split_bound_from(const Stats<CH>& a, const Stats<CH>& full, const SeedTables& T,
                                                    float rn_a = -1.f, float rn_b = -1.f)
{
    float cv1[10], cv2[10];
    covariance_of<CH>(cv1, a, rn_a < 0.f ? ispc_rcp(a.n, T) : rn_a);
    Stats<CH> b;
    #pragma unroll
    for (int i = 0; i < 10; i++) b.m[i] = full.m[i] - a.m[i];
    #pragma unroll
    for (int i = 0; i < 4; i++) b.s[i] = full.s[i] - a.s[i];
    b.n = full.n - a.n;
    covariance_of<CH>(cv2, b, rn_b < 0.f ? ispc_rcp(b.n, T) : rn_b);
    float bound = pca_residual<CH, FAST>(cv1, T);         // 0 + r1 of the reference: r1 = max(.., +0) is never -0
    bound += pca_residual<CH, FAST>(cv2, T);
    return f2i_x86(sqrtf(bound) * 256.0f);
}

template <int CH, bool FAST = false, class TX>
__device__ __forceinline__ int32_t split_bound(const TX& px, uint32_t mask0, const Stats<CH>& full, const SeedTables& T)
{
    Stats<CH> a;
    stats_of<CH>(a, px, mask0);
    return split_bound_from<CH, FAST>(a, full, T);
}

// ---- index selection (kernel.ispc:1133-1193) -------------------------------
// Interpolation weight of index q at BITS bits: round(64*q/(2^BITS-1)) = the format's weight tables.
template <int BITS>
__device__ __forceinline__ int32_t weight_of(int32_t q)
{
    constexpr int D = (1 << BITS) - 1;
    return (q * 128 + D) / (2 * D);
}

// For every texel: project on its subset's segment, try the two neighbouring indices, keep the better.
// Returns the summed (integer-truncated) squared error; indices packed 4 bits/texel into qb[2].
// HDR selects cvttps2dq semantics for the error truncation (BC6H errors overflow int; BC7's cannot).
template <int BITS, int CH, bool HDR, class TX>
__device__ __forceinline__ float select_indices(uint32_t (&qb)[2], const TX& px, const float (&ep)[3][2][4], uint32_t pattern)
{
    constexpr int LEVELS = 1 << BITS;
    float total = 0.f;
    qb[0] = qb[1] = 0u;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t j = (pattern >> (2 * k)) & 3u;
        float a[CH], b[CH];
        #pragma unroll
        for (int p = 0; p < CH; p++) {
            a[p] = (j == 0) ? ep[0][0][p] : ((j == 1) ? ep[1][0][p] : ep[2][0][p]);
            b[p] = (j == 0) ? ep[0][1][p] : ((j == 1) ? ep[1][1][p] : ep[2][1][p]);
        }
        float t[CH];
        #pragma unroll
        for (int p = 0; p < CH; p++) t[p] = px.get(p, k);
        float proj = 0.f, div = 0.f;
        #pragma unroll
        for (int p = 0; p < CH; p++) {
            proj += (t[p] - a[p]) * (b[p] - a[p]);
            div += sq(b[p] - a[p]);
        }
        proj = proj / div;                                   // IEEE divide (compound `/=` in the reference)
        int32_t q1 = f2i_x86(proj * (float)LEVELS + 0.5f);
        q1 = iclamp(q1, 1, LEVELS - 1);
        const float w0 = (float)weight_of<BITS>(q1 - 1), w1 = (float)weight_of<BITS>(q1);
        const float u0 = 64.0f - w0, u1 = 64.0f - w1;       // (64-w) is exact in int and in float
        float err0 = 0.f, err1 = 0.f;
        #pragma unroll
        for (int p = 0; p < CH; p++) {
            const float d0 = (float)f2i_x86((u0 * a[p] + w0 * b[p] + 32.0f) * 0.015625f);
            const float d1 = (float)f2i_x86((u1 * a[p] + w1 * b[p] + 32.0f) * 0.015625f);
            err0 += sq(d0 - t[p]);
            err1 += sq(d1 - t[p]);
        }
        const bool first = err0 < err1;
        const float e = first ? err0 : err1;
        const int32_t ei = HDR ? f2i_x86(e) : (int32_t)e;
        const uint32_t q = (uint32_t)(first ? q1 - 1 : q1);
        if (k < 8) qb[0] += q << (4 * k); else qb[1] += q << (4 * (k - 8));
        total += (float)ei;
    }
    return total;
}

// ---- least-squares endpoints for fixed indices (kernel.ispc:1198-1262) ------
template <int BITS, int CH, class TX>
__device__ __forceinline__ void refit_subset(float (&ep)[2][4], const TX& px, const uint32_t (&qb)[2], uint32_t mask, const SeedTables& T)
{
    constexpr float L1 = (float)((1 << BITS) - 1);
    float atb1[4] = {0.f, 0.f, 0.f, 0.f}, sum[4] = {0.f, 0.f, 0.f, 0.f};
    float sum_q = 0.f, sum_qq = 0.f, cnt = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        if ((mask >> k) & 1u) {
            const float q = (float)((k < 8 ? qb[0] >> (4 * k) : qb[1] >> (4 * (k - 8))) & 15u);
            const float x = L1 - q;
            sum_q += q;
            sum_qq += q * q;
            cnt += 1.0f;
            #pragma unroll
            for (int p = 0; p < CH; p++) { const float t = px.get(p, k); sum[p] += t; atb1[p] += x * t; }
        }
    }
    const float cxx = cnt * (L1 * L1) - (2.0f * L1) * sum_q + sum_qq;
    const float cyy = sum_qq;
    const float cxy = L1 * sum_q - sum_qq;
    const float det = cxx * cyy - cxy * cxy;
    const float scale = L1 * ispc_rcp(det, T);
    const bool flat = fabsf(det) < 0.001f;
    const float rcnt = ispc_rcp(cnt, T);
    #pragma unroll
    for (int p = 0; p < CH; p++) {
        const float atb2 = L1 * sum[p] - atb1[p];
        const float e0 = (atb1[p] * cyy - atb2 * cxy) * scale;
        const float e1 = (atb2 * cxx - atb1[p] * cxy) * scale;
        const float mean = sum[p] * rcnt;
        ep[0][p] = flat ? mean : e0;
        ep[1][p] = flat ? mean : e1;
    }
}

// ---- 128-bit block assembly --------------------------------------------------
// Fields are appended LSB first.  Header fields have lane-independent positions; the index fields are
// written full width and the implicit MSBs of the anchor texels are deleted afterwards (highest
// position first), like kernel.ispc:1746-1805 -- the deleted bits are zero by construction.
struct BlockBits {
    unsigned long long lo = 0, hi = 0;
    uint32_t ext = 0;          // bits 128.. while anchor MSBs are still in place
    __device__ __forceinline__ void put(int pos, int n, uint32_t v)
    {
        const unsigned long long x = (unsigned long long)v;
        if (pos < 64) {
            lo |= x << pos;
            if (pos + n > 64) hi |= x >> (64 - pos);
        } else if (pos < 128) {
            hi |= x << (pos - 64);
            if (pos + n > 128) ext |= (uint32_t)(x >> (128 - pos));
        } else {
            ext |= v << (pos - 128);
        }
    }
    // remove bit `at` (64 <= at < 128 + valid ext bits); everything above moves down by one
    __device__ __forceinline__ void drop_bit(int at)
    {
        const unsigned long long keep = (at >= 128) ? ~0ull : ((1ull << (at - 64)) - 1ull);
        const unsigned long long shifted = (hi >> 1) | ((unsigned long long)(ext & 1u) << 63);
        if (at < 128) { hi = (hi & keep) | (shifted & ~keep); ext >>= 1; }
        else          { const uint32_t k2 = (1u << (at - 128)) - 1u; ext = (ext & k2) | ((ext >> 1) & ~k2); }
    }
};

} // namespace itw
