// convert.hip -- Photoshop-buffer -> encoder-surface conversions on the GPU (include/itw_dispatch.h): the step
// before padding and the ABI in the reference's save path (IntelPlugin.cpp:741-810 ConvertToBCFrom8/16/32Bit,
// :291-366 ConvertToBC6From8/16/32Bit; helpers IntelPlugin.h:31-96).  One pixel per lane, 1-16 B in, 4 / 8 B out:
// a stream, HBM bound.
//   8 -> 8  : copy                                   16 -> 8 : v > 32768 ? 255 : (v*255) >> 15   (= FloatToByte(v/32768.0))
//   32 -> 8 : FloatToByte(pow(v, 1/2.2)) in double   (pow is not bit-pinned across platforms: +-1 code vs the CPU)
//   8 -> 16F: half(v / 255.f)    16 -> 16F: half((float)(v / 32768.0))    32 -> 16F: half(v)   (round to nearest even)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include "../../include/itw_dispatch.h"
#include "../../include/itw_amd.h"

namespace {

__device__ __forceinline__ uint32_t float_to_byte(double v)                 // IntelPlugin.h:41-48
{
    if (v > 1) return 255u;
    if (v < 0) return 0u;
    return (uint32_t)(v * 255) & 255u;
}

__device__ __forceinline__ uint32_t to_half_bits(float v)
{
    return (uint32_t)__half_as_ushort(__float2half_rn(v));
}

template <int DEPTH>
__device__ __forceinline__ uint32_t to8(const void* src, int64_t idx, bool gamma)
{
    if (DEPTH == 8) return ((const uint8_t*)src)[idx];
    if (DEPTH == 16) { const uint32_t v = ((const uint16_t*)src)[idx]; return v > 32768u ? 255u : (v * 255u) >> 15; }
    double v = (double)((const float*)src)[idx];
    if (gamma) v = pow(v, 1 / 2.2);
    return float_to_byte(v);
}

template <int DEPTH>
__device__ __forceinline__ uint32_t to16f(const void* src, int64_t idx)
{
    if (DEPTH == 8) return to_half_bits((float)((const uint8_t*)src)[idx] / 255.f);
    if (DEPTH == 16) return to_half_bits((float)((double)((const uint16_t*)src)[idx] / 32768.0));
    return to_half_bits(((const float*)src)[idx]);
}

template <int DEPTH, bool HALF>
__global__ void __launch_bounds__(256)
convert_kernel(const void* __restrict__ src, int planes, bool has_alpha, bool gamma, int64_t npix, uint32_t* __restrict__ dst)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= npix) return;
    uint32_t c[4] = {0u, 0u, 0u, HALF ? 0x3c00u : 255u};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const bool present = (k < 3) ? (k < planes) : has_alpha;
        if (!present) continue;
        const int64_t idx = i * planes + ((HALF && k == 3 && DEPTH == 32) ? 2 : k);      // IntelPlugin.cpp:361 reads plane 2
        c[k] = HALF ? to16f<DEPTH>(src, idx) : to8<DEPTH>(src, idx, gamma);
    }
    if (HALF) { dst[2 * i] = c[0] | (c[1] << 16); dst[2 * i + 1] = c[2] | (c[3] << 16); }
    else      dst[i] = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
}

int launch(const void* src, int depth, int planes, int has_alpha, int gamma, int width, int height, void* dst, bool half)
{
    if ((depth != 8 && depth != 16 && depth != 32) || planes < 1 || planes > 4 || width <= 0 || height <= 0) return -1;
    if (has_alpha && planes < ((half && depth == 32) ? 3 : 4)) return -1;
    const int64_t n = (int64_t)width * height;
    hipStream_t st = (hipStream_t)itwGetStream();
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    uint32_t* d = (uint32_t*)dst;
#define ITW_CONV(D, H) hipLaunchKernelGGL((convert_kernel<D, H>), grid, blk, 0, st, src, planes, has_alpha != 0, gamma != 0, n, d)
    if (half) { if (depth == 8) ITW_CONV(8, true); else if (depth == 16) ITW_CONV(16, true); else ITW_CONV(32, true); }
    else      { if (depth == 8) ITW_CONV(8, false); else if (depth == 16) ITW_CONV(16, false); else ITW_CONV(32, false); }
#undef ITW_CONV
    if (hipGetLastError() != hipSuccess) { std::fprintf(stderr, "itwConvert: launch failed\n"); std::abort(); }
    return 0;
}

} // namespace

extern "C" int itwConvertToRGBA8Device(const void* src, int depth, int planes, int has_alpha, int gamma_correct, int width, int height, uint8_t* dst)
{
    return launch(src, depth, planes, has_alpha, gamma_correct, width, height, dst, false);
}

extern "C" int itwConvertToRGBA16FDevice(const void* src, int depth, int planes, int has_alpha, int width, int height, uint16_t* dst)
{
    return launch(src, depth, planes, has_alpha, 0, width, height, dst, true);
}
