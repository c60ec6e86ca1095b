// x86_math.hpp -- device-side pinned arithmetic for the BCn kernels (gfx950).
//
// The reference's shipped object code (ispc --opt=fast-math, sse/avx targets;
// IntelTextureWorks.vcxproj:388) computes every float divide as x*rcp(y) where
// rcp()/rsqrt() are Intel's RCPPS/RSQRTPS table seeds refined by one
// Newton-Raphson step, truncates with cvttps2dq and takes min/max with
// minps/maxps.  To emit the same blocks, the kernels never use the GPU's own
// v_rcp_f32 / v_rsq_f32 / v_cvt_i32_f32 / v_min_f32 semantics for those steps;
// they go through the helpers below.  Translation units are compiled with
// -ffp-contract=off so no multiply-add is ever fused.
//
// Seeds: two 2048-entry tables packed to 16 bit (x86_luts_packed.h, generated
// and exhaustively verified by tools/extract_x86_luts.c).  Kernels read them
// through a pointer so a workgroup can serve them from LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace itw {

#define X86_LUT_QUAL __device__ const
#include "x86_luts_packed.h"
#undef X86_LUT_QUAL

struct SeedTables {
    const unsigned short* rcp;     // [2048] indexed by mantissa[22:12]
    const unsigned short* rsqrt;   // [2048] indexed by {exponent parity, mantissa[22:13]}
};

__device__ __forceinline__ SeedTables global_seed_tables()
{
    return SeedTables{X86_RCP_SEED16, X86_RSQRT_SEED16};
}

// Cooperative copy of both tables (8 KiB) into LDS; `lds` must hold 4096 ushorts.
__device__ __forceinline__ SeedTables stage_seed_tables(unsigned short* lds, int tid, int nthreads)
{
    const uint32_t* s0 = reinterpret_cast<const uint32_t*>(X86_RCP_SEED16);
    const uint32_t* s1 = reinterpret_cast<const uint32_t*>(X86_RSQRT_SEED16);
    uint32_t* d = reinterpret_cast<uint32_t*>(lds);
    for (int i = tid; i < 1024; i += nthreads) { d[i] = s0[i]; d[1024 + i] = s1[i]; }
    return SeedTables{lds, lds + 2048};
}

// RCPPS.  Model proven equal to the instruction for all 2^32 inputs.
__device__ __forceinline__ float x86_rcpps(float v, const unsigned short* lut)
{
    const uint32_t x = __float_as_uint(v);
    const uint32_t s = x & 0x80000000u, e = (x >> 23) & 255u, m = x & 0x7fffffu;
    const uint32_t t = 0x3f000000u | ((uint32_t)lut[m >> 12] << 11);
    const int32_t re = (int32_t)(t >> 23) + 127 - (int32_t)e;
    uint32_t r = s | ((uint32_t)re << 23) | (t & 0x7fffffu);
    r = (re <= 0) ? s : r;                                   // denormal result: flushed
    r = (e == 0u) ? (s | 0x7f800000u) : r;                   // zero / denormal operand
    r = (e == 255u) ? (m ? (x | 0x00400000u) : s) : r;       // NaN / inf
    return __uint_as_float(r);
}

// RSQRTPS.
__device__ __forceinline__ float x86_rsqrtps(float v, const unsigned short* lut)
{
    const uint32_t x = __float_as_uint(v);
    const uint32_t s = x & 0x80000000u, e = (x >> 23) & 255u, m = x & 0x7fffffu;
    const uint32_t odd = (~e) & 1u;                           // exponent field 127 (odd) = even power of two
    const uint32_t t = 0x3f000000u | ((uint32_t)lut[(odd << 10) | (m >> 13)] << 11);
    const int32_t k = ((int32_t)e - (int32_t)(127u + odd)) >> 1; // exact halving (difference is even)
    uint32_t r = t - ((uint32_t)k << 23);
    r = (e == 255u) ? 0u : r;                                 // +inf -> +0
    r = s ? 0xffc00000u : r;                                  // negative -> default NaN
    r = (e == 0u) ? (s | 0x7f800000u) : r;                    // +-0 / denormal -> +-inf
    r = (e == 255u && m) ? (x | 0x00400000u) : r;             // NaN
    return __uint_as_float(r);
}

// ISPC stdlib rcp(): r*(2 - v*r), three separately rounded operations.
__device__ __forceinline__ float ispc_rcp(float v, const SeedTables& T)
{
    const float r = x86_rcpps(v, T.rcp);
    float t = v * r;
    t = 2.0f - t;
    return r * t;
}

// ISPC stdlib rsqrt(): 0.5*(is*(3 - (v*is)*is)).
__device__ __forceinline__ float ispc_rsqrt(float v, const SeedTables& T)
{
    const float is = x86_rsqrtps(v, T.rsqrt);
    float a = v * is;
    a = a * is;
    a = 3.0f - a;
    a = is * a;
    return 0.5f * a;
}

// cvttps2dq: truncate, everything unrepresentable (incl. NaN) -> INT_MIN.
// v_cvt_i32_f32 saturates and maps NaN to 0, hence the explicit range test.
__device__ __forceinline__ int32_t f2i_x86(float f)
{
    const bool ok = (f >= -2147483648.0f) && (f < 2147483648.0f);
    return ok ? (int32_t)f : (int32_t)0x80000000;
}

// minps / maxps: second operand wins when unordered.
__device__ __forceinline__ float fmin_x86(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float fmax_x86(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float fclamp_x86(float v, float lo, float hi) { return fmin_x86(fmax_x86(v, lo), hi); }

__device__ __forceinline__ int32_t iclamp(int32_t v, int32_t lo, int32_t hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ float sq(float v) { return v * v; }

} // namespace itw
