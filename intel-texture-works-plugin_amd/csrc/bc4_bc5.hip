// bc4_bc5.hip -- BC4_UNORM / BC5_UNORM encoders for gfx950.
//
// These two formats are the only ones the plugin does not send through kernel.ispc: IntelPlugin.cpp:271-273 hands the
// RGBA8 scratch image to DirectXTex (DirectX::Compress, TEX_COMPRESS_DEFAULT).  This file is the device replacement of
// that call: same block walk and partial-block rule (DirectXTexCompress.cpp:105-183), same endpoint optimiser
// (BC.h:727-856 OptimizeAlpha<false>, BC4BC5.cpp:186-238 FindEndPointsBC4U) and index choice (BC4BC5.cpp:314-337),
// evaluated in IEEE fp32 exactly as written, one rounding per operation (the library is built with -ffp-contract=off).
//
// Mapping: one lane per 8-byte channel block (BC5: lanes 2b and 2b+1 encode R and G of block b), so a wavefront writes
// 512 contiguous bytes; each lane keeps its 16 texels in registers and runs the (at most eight) Newton iterations
// with a per-lane exit.  The ramp weights k/5 and k/7 are compile-time fp32 quotients in the reference; k * (1/7)
// is not bit-equal to k/7 for k = 3 and 6, so they come from a 14-entry LDS table instead of arithmetic.
// HBM-bound in principle (64 B in, 8 or 16 B out per block); in practice VALU-bound like BC1 (~3 k fp32 ops/channel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "kernels.hpp"

namespace itw {
namespace {

// k/7 (k = 0..7) then k/5 (k = 0..5): pD8, pD6 of BC.h:729-732; pC is the same table read backwards.
__device__ const float RAMP_WEIGHTS[14] = {
    0.0f / 7.0f, 1.0f / 7.0f, 2.0f / 7.0f, 3.0f / 7.0f, 4.0f / 7.0f, 5.0f / 7.0f, 6.0f / 7.0f, 7.0f / 7.0f,
    0.0f / 5.0f, 1.0f / 5.0f, 2.0f / 5.0f, 3.0f / 5.0f, 4.0f / 5.0f, 5.0f / 5.0f };

// OptimizeAlpha<false> (BC.h:727-856).  STEPS = 8: plain ramp; 6: ramp plus the exact codes 0 and 1.
template <int STEPS>
__device__ __forceinline__ void optimize_ramp(float& out_x, float& out_y, const float (&t)[16], const float* tab)
{
    const float* w = tab + (STEPS == 8 ? 0 : 8);
    float fx = 1.0f, fy = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        if (STEPS == 8) {
            if (t[i] < fx) fx = t[i];
            if (t[i] > fy) fy = t[i];
        } else {
            if (t[i] < fx && t[i] > 0.0f) fx = t[i];
            if (t[i] > fy && t[i] < 1.0f) fy = t[i];
        }
    }
    if (STEPS == 6 && fx == fy) fy = 1.0f;
    const float fsteps = (float)(STEPS - 1);
#pragma unroll 1
    for (int it = 0; it < 8; it++) {
        if ((fy - fx) < (1.0f / 256.0f)) break;
        const float scale = fsteps / (fy - fx);
        float dx = 0.0f, dy = 0.0f, d2x = 0.0f, d2y = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float dot = (t[i] - fx) * scale;
            int s;
            if (dot <= 0.0f) s = (STEPS == 6 && t[i] <= fx * 0.5f) ? 6 : 0;
            else if (dot >= fsteps) s = (STEPS == 6 && t[i] >= (fy + 1.0f) * 0.5f) ? 7 : (STEPS - 1);
            else s = (int)(dot + 0.5f);
            if (s < STEPS) {
                const float c = w[STEPS - 1 - s], d = w[s];
                const float diff = (c * fx + d * fy) - t[i];
                dx += c * diff;
                d2x += c * c;
                dy += d * diff;
                d2y += d * d;
            }
        }
        if (d2x > 0.0f) fx -= dx / d2x;
        if (d2y > 0.0f) fy -= dy / d2y;
        if (fx > fy) { const float f = fx; fx = fy; fy = f; }
        if ((dx * dx < (1.0f / 64.0f)) && (dy * dy < (1.0f / 64.0f))) break;
    }
    out_x = (fx < 0.0f) ? 0.0f : (fx > 1.0f) ? 1.0f : fx;
    out_y = (fy < 0.0f) ? 0.0f : (fy > 1.0f) ? 1.0f : fy;
}

// One channel of one block -> 8 bytes (D3DXEncodeBC4U, BC4BC5.cpp:403-421).
__device__ __forceinline__ uint2 encode_channel(const float (&t)[16], const float* tab)
{
    float bmin = t[0], bmax = t[0];
#pragma unroll
    for (int i = 1; i < 16; i++) { bmin = t[i] < bmin ? t[i] : bmin; bmax = t[i] > bmax ? t[i] : bmax; }
    uint32_t r0, r1;
    float fs, fe;
    if (!(0.0f == bmin || 1.0f == bmax)) {                       // BC4BC5.cpp:213-237
        optimize_ramp<8>(fs, fe, t, tab);
        r0 = (uint32_t)(fe * 255.0f);
        r1 = (uint32_t)(fs * 255.0f);
    } else {
        optimize_ramp<6>(fs, fe, t, tab);
        r1 = (uint32_t)(fe * 255.0f);
        r0 = (uint32_t)(fs * 255.0f);
    }
    // the eight decoded values, BC4_UNORM::DecodeFromIndex (BC4BC5.cpp:50-72)
    const float f0 = (float)r0 / 255.0f, f1 = (float)r1 / 255.0f;
    float g[8];
    g[0] = f0; g[1] = f1;
    const bool eight = r0 > r1;
#pragma unroll
    for (int k = 1; k <= 6; k++) {
        const float v8 = (f0 * (float)(7 - k) + f1 * (float)k) / 7.0f;
        const float v6 = (k <= 4) ? (f0 * (float)(5 - k) + f1 * (float)k) / 5.0f : (k == 5 ? 0.0f : 1.0f);
        g[k + 1] = eight ? v8 : v6;
    }
    // FindClosestUNORM (BC4BC5.cpp:314-337): first index with the strictly smallest |g - t|
    uint64_t idx = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        uint32_t best = 0;
        float best_delta = 100000.0f;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const float d = fabsf(g[k] - t[i]);
            if (d < best_delta) { best = k; best_delta = d; }
        }
        idx |= (uint64_t)best << (3 * i);
    }
    const uint64_t data = (uint64_t)r0 | ((uint64_t)r1 << 8) | (idx << 16);
    return make_uint2((uint32_t)data, (uint32_t)(data >> 32));
}

template <int NCH, bool VEC16>
__global__ void __launch_bounds__(256)
bc45_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t width, int32_t height, int32_t blocks_x,
            int32_t nlanes, uint8_t* __restrict__ dst)
{
    __shared__ float s_tab[16];
    if (threadIdx.x < 14) s_tab[threadIdx.x] = RAMP_WEIGHTS[threadIdx.x];
    __syncthreads();
    const int32_t lane = blockIdx.x * 256 + threadIdx.x;
    if (lane >= nlanes) return;
    const int32_t b = (NCH == 2) ? (lane >> 1) : lane;
    const uint32_t shift = (NCH == 2) ? (uint32_t)(lane & 1) * 8u : 0u;
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    const int32_t pw = min(4, width - 4 * xx), ph = min(4, height - 4 * yy);
    const uint8_t* p = src + (int64_t)yy * 4 * stride + (int64_t)xx * 16;
    const float scale = 1.0f / 255.0f;                           // XMLoadUByteN4's SSE path: integer * (1/255)

    float t[16];
    if (pw == 4 && ph == 4) {
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
            for (int x = 0; x < 4; x++) t[y * 4 + x] = (float)((w[x] >> shift) & 255u) * scale;
        }
    } else {
        // partial block: missing columns / rows repeat source column / row {0,0,0,1}[i], itself wrapped to 0 when
        // that one is missing too (DirectXTexCompress.cpp:140-168 applied in its own order)
#pragma unroll
        for (int y = 0; y < 4; y++) {
            int sy = y < ph ? y : (y == 3 ? 1 : 0);
            if (sy >= ph) sy = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                int sx = x < pw ? x : (x == 3 ? 1 : 0);
                if (sx >= pw) sx = 0;
                const uint32_t w = *reinterpret_cast<const uint32_t*>(p + sy * stride + sx * 4);
                t[y * 4 + x] = (float)((w >> shift) & 255u) * scale;
            }
        }
    }
    const uint2 o = encode_channel(t, s_tab);
    *reinterpret_cast<uint2*>(dst + (int64_t)lane * 8) = o;
}

template <int NCH>
void launch_bc45(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st)
{
    if (width <= 0 || height <= 0) return;
    const int bx = (width + 3) / 4, by = (height + 3) / 4;       // DirectXTex keeps partial blocks
    const int64_t n = (int64_t)bx * by * NCH;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 15) == 0;
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    if (vec) hipLaunchKernelGGL((bc45_kernel<NCH, true>),  grid, blk, 0, st, src, stride, width, height, bx, (int32_t)n, dst);
    else     hipLaunchKernelGGL((bc45_kernel<NCH, false>), grid, blk, 0, st, src, stride, width, height, bx, (int32_t)n, dst);
}

} // namespace

void launch_bc4(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st) { launch_bc45<1>(src, stride, width, height, dst, st); }
void launch_bc5(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st) { launch_bc45<2>(src, stride, width, height, dst, st); }

} // namespace itw
