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
// ~1.5 k VALU operations per block, so it sits near the HBM/VALU ridge; there is
// no reuse to stage in LDS and nothing matrix shaped for MFMA.
//
// Arithmetic is the pinned x86 model of x86_math.hpp; float sums run serially in
// texel order k = 0..15 inside the lane, exactly like one ISPC program instance.
#include "x86_math.hpp"
#include "kernels.hpp"

namespace itw {

// 8-bit -> 5/6-bit with rounding, (t + (t>>8)) >> 8 form.   [kernel.ispc:234-248]
__device__ __forceinline__ int32_t scale8(int32_t a, int32_t b)
{
    const int32_t t = a * b + 128;
    return (t + (t >> 8)) >> 8;
}

// inputs are clamped to [0,255] by the callers (fclamp_x86 maps NaN to 0), so the plain conversion is cvttps2dq
__device__ __forceinline__ int32_t pack565(float cr, float cg, float cb)
{
    return ((scale8(cvt_i32_sat(cr), 31) << 11) + (scale8(cvt_i32_sat(cg), 63) << 5) + scale8(cvt_i32_sat(cb), 31)) & 0xffff;
}

__device__ __forceinline__ void unpack565(int32_t p, float c[3])        // [kernel.ispc:250-259]
{
    const int32_t b5 = p & 31, g6 = (p >> 5) & 63, r5 = (p >> 11) & 31;
    c[0] = (float)((r5 << 3) + (r5 >> 2));
    c[1] = (float)((g6 << 2) + (g6 >> 4));
    c[2] = (float)((b5 << 3) + (b5 >> 2));
}

// Project the 16 texels on the endpoint segment and emit linear 2-bit indices.
// [kernel.ispc:308-344]
__device__ __forceinline__ uint32_t project_indices(const float (&px)[3][16], int32_t p0, int32_t p1, const SeedTables& T)
{
    float c0[3], c1[3], dir[3];
    unpack565(p0, c0);
    unpack565(p1, c1);
    for (int p = 0; p < 3; p++) dir[p] = c1[p] - c0[p];

    float sq_norm = 0.f;
    for (int p = 0; p < 3; p++) sq_norm += sq(dir[p]);
    const float rs3 = ispc_rcp(sq_norm, T) * 3.0f;
    for (int p = 0; p < 3; p++) dir[p] *= rs3;

    float bias = 0.5f;
    for (int p = 0; p < 3; p++) bias -= c0[p] * dir[p];

    uint32_t bits = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        float dot = 0.f;
        for (int p = 0; p < 3; p++) dot += px[p][k] * dir[p];
        // finite and small (|dir| <= 3*255, texels <= 255) or NaN when p0 == p1 (rcp(0)): NaN -> 0 on both routes after the clamp
        const int32_t q = iclamp(cvt_i32_sat(dot + bias), 0, 3);
        bits += (uint32_t)q << (2 * k);           // q*4^k, no carries: q < 4
    }
    return bits;
}

// Least-squares endpoint update for fixed indices.               [kernel.ispc:419-480]
__device__ __forceinline__ void refit_endpoints(int32_t pe[2], const float (&px)[3][16], uint32_t bits,
                                                const float dc[3], const SeedTables& T)
{
    float c0[3], c1[3];
    if ((bits ^ (bits * 4u)) < 4u) {
        for (int p = 0; p < 3; p++) { c0[p] = dc[p]; c1[p] = dc[p]; }
    } else {
        float atb1[3] = {0.f, 0.f, 0.f};
        float sum_q = 0.f, sum_qq = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const float q = (float)(int32_t)((bits >> (2 * k)) & 3u);
            const float x = 3.0f - q;
            sum_q += q;
            sum_qq += q * q;
            for (int p = 0; p < 3; p++) atb1[p] += x * px[p][k];
        }
        const float cxx = 144.0f - 6.0f * sum_q + sum_qq;
        const float cyy = sum_qq;
        const float cxy = 3.0f * sum_q - sum_qq;
        const float scale = 3.0f * ispc_rcp(cxx * cyy - cxy * cxy, T);
        for (int p = 0; p < 3; p++) {
            const float sum = dc[p] * 16.0f;
            const float atb2 = 3.0f * sum - atb1[p];
            c0[p] = fclamp_num((atb1[p] * cyy - atb2 * cxy) * scale, 0.f, 255.f);
            c1[p] = fclamp_num((atb2 * cxx - atb1[p] * cxy) * scale, 0.f, 255.f);
        }
    }
    pe[0] = pack565(c0[0], c0[1], c0[2]);
    pe[1] = pack565(c1[0], c1[1], c1[2]);
}

// Colour part: PCA axis by power iteration, endpoint pick, one refit pass.
// [kernel.ispc:494-533]
__device__ __forceinline__ void encode_color(const float (&px)[3][16], uint32_t out[2], const SeedTables& T)
{
    float dc[3];
    for (int p = 0; p < 3; p++) {
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < 16; k++) acc += px[p][k];
        dc[p] = acc * 0.0625f;
    }

    // packed symmetric covariance  [rr rg rb gg gb bb]          [kernel.ispc:377-417]
    float cv[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const float r = px[0][k] - dc[0], g = px[1][k] - dc[1], b = px[2][k] - dc[2];
        cv[0] += r * r; cv[1] += r * g; cv[2] += r * b;
        cv[3] += g * g; cv[4] += g * b; cv[5] += b * b;
    }
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
            const float rn = ispc_rsqrt(n, T);
            v[0] *= rn; v[1] *= rn; v[2] *= rn;
        }
    }

    // extreme projections -> endpoints                           [kernel.ispc:274-306]
    float lo = 65536.0f, hi = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        float dot = 0.f;
        for (int p = 0; p < 3; p++) dot += (px[p][k] - dc[p]) * v[p];
        lo = fmin_x86(lo, dot);
        hi = fmax_x86(hi, dot);
    }
    if (hi - lo < 1.0f) { lo -= 0.5f; hi += 0.5f; }

    float nsq = 0.f;
    for (int p = 0; p < 3; p++) nsq += v[p] * v[p];
    const float rn = ispc_rcp(nsq, T);

    float e0[3], e1[3];
    for (int p = 0; p < 3; p++) {
        e0[p] = fclamp_num(dc[p] + lo * rn * v[p], 0.f, 255.f);
        e1[p] = fclamp_num(dc[p] + hi * rn * v[p], 0.f, 255.f);
    }

    int32_t pe[2];
    pe[0] = pack565(e0[0], e0[1], e0[2]);
    pe[1] = pack565(e1[0], e1[1], e1[2]);
    if (pe[0] < pe[1]) { const int32_t t = pe[0]; pe[0] = pe[1]; pe[1] = t; }   // keep 4-colour mode
    uint32_t idx = project_indices(px, pe[0], pe[1], T);

    refit_endpoints(pe, px, idx, dc, T);
    if (pe[0] < pe[1]) { const int32_t t = pe[0]; pe[0] = pe[1]; pe[1] = t; }
    idx = project_indices(px, pe[0], pe[1], T);

    out[0] = ((uint32_t)pe[1] << 16) + (uint32_t)pe[0];
    // linear order {0,1,2,3} -> BC1 order {0,2,3,1}              [kernel.ispc:482-492]
    const uint32_t lo_bits = idx & 0x55555555u, hi_bits = idx & 0xAAAAAAAAu;
    out[1] = (hi_bits >> 1) + (hi_bits ^ (lo_bits << 1));
}

// Alpha part of BC3: min/max endpoints, 8-level ramp.            [kernel.ispc:535-571]
__device__ __forceinline__ void encode_alpha(const float (&a)[16], uint32_t out[2], const SeedTables& T)
{
    float lo = 255.f, hi = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) { lo = __builtin_fminf(lo, a[k]); hi = __builtin_fmaxf(hi, a[k]); }   // minps / maxps of ordinary numbers (bytes)
    if (lo == hi) hi = lo + 0.1f;
    const float scale = 7.0f * ispc_rcp(hi - lo, T);

    uint32_t q0 = 0, q1 = 0;      // 8 x 3 bits each
#pragma unroll
    for (int k = 0; k < 16; k++) {
        int32_t q = 7 - iclamp(cvt_i32_sat((a[k] - lo) * scale + 0.5f), 0, 7);      // hi - lo >= 0.1: finite, in [0.5, 7.6]
        q = (q > 0) ? q + 1 : q;   // DXT5 order: 0 = alpha0(max), 1 = alpha1(min), 2.. ramp
        q = (q == 8) ? 1 : q;
        if (k < 8) q0 |= (uint32_t)q << (k * 3); else q1 |= (uint32_t)q << ((k - 8) * 3);
    }
    out[0] = (uint32_t)(iclamp(cvt_i32_sat(lo), 0, 255) * 256 + iclamp(cvt_i32_sat(hi), 0, 255)) | (q0 << 16);
    out[1] = (q0 >> 16) | (q1 << 8);
}

template <bool BC3, bool VEC16>
__global__ void __launch_bounds__(256)
bc13_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t blocks_x, int32_t nblocks, uint8_t* __restrict__ dst)
{
    const int32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    const SeedTables T = global_seed_tables();

    float px[3][16];
    float al[16];
    const uint8_t* p = src + (int64_t)yy * 4 * stride + (int64_t)xx * 16;
#pragma unroll
    for (int y = 0; y < 4; y++) {
        uint32_t w[4];
        if (VEC16) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + y * stride);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            const uint32_t* q = reinterpret_cast<const uint32_t*>(p + y * stride);
            w[0] = q[0]; w[1] = q[1]; w[2] = q[2]; w[3] = q[3];
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
            px[0][y * 4 + x] = (float)(w[x] & 255u);
            px[1][y * 4 + x] = (float)((w[x] >> 8) & 255u);
            px[2][y * 4 + x] = (float)((w[x] >> 16) & 255u);
            if (BC3) al[y * 4 + x] = (float)(w[x] >> 24);
        }
    }

    if (BC3) {
        uint32_t o[4];
        encode_alpha(al, &o[0], T);
        encode_color(px, &o[2], T);
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (int64_t)b * 16);
        if (VEC16) *reinterpret_cast<uint4*>(d) = make_uint4(o[0], o[1], o[2], o[3]);
        else { d[0] = o[0]; d[1] = o[1]; d[2] = o[2]; d[3] = o[3]; }
    } else {
        uint32_t o[2];
        encode_color(px, o, T);
        uint32_t* d = reinterpret_cast<uint32_t*>(dst + (int64_t)b * 8);
        if (VEC16) *reinterpret_cast<uint2*>(d) = make_uint2(o[0], o[1]);
        else { d[0] = o[0]; d[1] = o[1]; }
    }
}

// VEC16 requires: src base and stride multiples of 16, dst multiple of 16 (BC3) / 8 (BC1).
void launch_bc1(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st)
{
    const int bx = width / 4, by = height / 4;
    const int64_t n = (int64_t)bx * by;
    if (n <= 0) return;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 15) == 0 && (reinterpret_cast<uintptr_t>(dst) & 7) == 0;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (vec) hipLaunchKernelGGL((bc13_kernel<false, true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
    else     hipLaunchKernelGGL((bc13_kernel<false, false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
}

void launch_bc3(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st)
{
    const int bx = width / 4, by = height / 4;
    const int64_t n = (int64_t)bx * by;
    if (n <= 0) return;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride | reinterpret_cast<uintptr_t>(dst)) & 15) == 0;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (vec) hipLaunchKernelGGL((bc13_kernel<true, true>),  grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
    else     hipLaunchKernelGGL((bc13_kernel<true, false>), grid, blk, 0, st, src, stride, bx, (int32_t)n, dst);
}

} // namespace itw
