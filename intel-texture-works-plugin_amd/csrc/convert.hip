// convert.hip -- Photoshop-buffer -> encoder-surface conversions on the GPU (include/itw_dispatch.h): the step
// before padding and the ABI in the reference's save path (IntelPlugin.cpp:741-810 ConvertToBCFrom8/16/32Bit,
// :291-366 ConvertToBC6From8/16/32Bit; helpers IntelPlugin.h:31-96).  One pixel per lane, 1-16 B in, 4 / 8 B out:
// a stream, HBM bound.
//   8 -> 8  : copy                                   16 -> 8 : v > 32768 ? 255 : (v*255) >> 15   (= FloatToByte(v/32768.0))
//   32 -> 8 : FloatToByte(v) in double; with gamma FloatToByte(pow(v, 1/2.2)) -- evaluated as "how many of the 255 code thresholds are <= v"
//             (gamma_thresholds.h: the thresholds of the reference's own function compiled in the build container), eight comparisons, no
//             pow: bit-exact with that function, where a device-library pow was +-1 code
//   8 -> 16F: half(v / 255.f)    16 -> 16F: half((float)(v / 32768.0))    32 -> 16F: half(v)   (round to nearest even)
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include "../../include/itw_dispatch.h"
#include "../../include/itw_amd.h"
#include "gamma_thresholds.h"

namespace {

__device__ const uint32_t GAMMA_THR[256] = {
#define ITW_T(i) itw::GAMMA_THR_BITS[i]
#define ITW_T8(i) ITW_T(i), ITW_T(i + 1), ITW_T(i + 2), ITW_T(i + 3), ITW_T(i + 4), ITW_T(i + 5), ITW_T(i + 6), ITW_T(i + 7)
#define ITW_T64(i) ITW_T8(i), ITW_T8(i + 8), ITW_T8(i + 16), ITW_T8(i + 24), ITW_T8(i + 32), ITW_T8(i + 40), ITW_T8(i + 48), ITW_T8(i + 56)
    ITW_T64(0), ITW_T64(64), ITW_T64(128), ITW_T64(192)
#undef ITW_T64
#undef ITW_T8
#undef ITW_T
};

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
    const float f = ((const float*)src)[idx];
    if (gamma) {
        // ConvertTo8Bit(v, true) is non-decreasing in v: its value is the number of code thresholds <= v.  NaN and negative finite v compare below
        // every threshold -> 0, like (unsigned char)(NaN * 255) of the reference's x86 code; v > 1 passes all 255.
        if (f == -__builtin_inff()) return 255u;      // pow(-inf, y > 0, not an odd integer) = +inf (C99 F.9.4.4): the one negative input above 1
        uint32_t lo = 0;
#pragma unroll
        for (uint32_t step = 128; step >= 1; step >>= 1)
            if (f >= __uint_as_float(GAMMA_THR[lo + step])) lo += step;           // lo + step <= 255
        return lo;
    }
    return float_to_byte((double)f);
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
