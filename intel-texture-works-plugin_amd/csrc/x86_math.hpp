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
    const uint32_t* rsqrt32;       // [2048] LDS, fast path (see x86_rsqrtps_fast); null when not staged
};

__device__ __forceinline__ SeedTables global_seed_tables()
{
    return SeedTables{X86_RCP_SEED16, X86_RSQRT_SEED16, nullptr};
}

// Cooperative copy of both tables (8 KiB) into LDS; `lds` must hold 4096 ushorts.
__device__ __forceinline__ SeedTables stage_seed_tables(unsigned short* lds, int tid, int nthreads)
{
    const uint32_t* s0 = reinterpret_cast<const uint32_t*>(X86_RCP_SEED16);
    const uint32_t* s1 = reinterpret_cast<const uint32_t*>(X86_RSQRT_SEED16);
    uint32_t* d = reinterpret_cast<uint32_t*>(lds);
    for (int i = tid; i < 1024; i += nthreads) { d[i] = s0[i]; d[1024 + i] = s1[i]; }
    return SeedTables{lds, lds + 2048, nullptr};
}

// LDS staging for kernels that normalise a lot (BC7): the RCPPS table as is (4 KiB at lds16) and the RSQRTPS table
// expanded to ready-to-use words (8 KiB at lds32): entry i, indexed by bits [23:13] of the operand (exponent LSB,
// top 10 mantissa bits), holds seed mantissa | 0.5's exponent, plus 64 << 23 so that the result's exponent is one
// subtraction away.  `lds16` must hold 2048 ushorts, `lds32` 2048 words.
__device__ __forceinline__ SeedTables stage_seed_tables_fast(unsigned short* lds16, uint32_t* lds32, int tid, int nthreads)
{
    const uint32_t* s0 = reinterpret_cast<const uint32_t*>(X86_RCP_SEED16);
    uint32_t* d = reinterpret_cast<uint32_t*>(lds16);
    for (int i = tid; i < 1024; i += nthreads) d[i] = s0[i];
    for (int i = tid; i < 2048; i += nthreads)
        lds32[i] = (0x3f000000u | ((uint32_t)X86_RSQRT_SEED16[i ^ 0x400] << 11)) + 0x20000000u;
    return SeedTables{lds16, X86_RSQRT_SEED16, lds32};
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

// RSQRTPS of a positive normal operand in 6 VALU operations + one LDS read; everything else (zero, denormal,
// negative, inf, NaN: never produced by well-formed blocks) takes the general model above in a divergent branch.
// With e the exponent field and h = ((e + 1) >> 1) - 64:  result = seed_word(i) - (h << 23)  (same as x86_rsqrtps:
// odd = ~e & 1, k = (e - 127 - odd) >> 1 = h), and ((x + 0x00800000) >> 1) & 0x7f800000 = ((e + 1) >> 1) << 23.
__device__ __forceinline__ float x86_rsqrtps_fast(float v, const SeedTables& T)
{
    const uint32_t x = __float_as_uint(v);
    if (__builtin_expect((x - 0x00800000u) >= 0x7f000000u, 0)) return x86_rsqrtps(v, T.rsqrt);
    const uint32_t t = T.rsqrt32[(x >> 13) & 0x7ffu];
    return __uint_as_float(t - (((x + 0x00800000u) >> 1) & 0x7f800000u));
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
template <bool FAST = false>
__device__ __forceinline__ float ispc_rsqrt(float v, const SeedTables& T)
{
    const float is = FAST ? x86_rsqrtps_fast(v, T) : x86_rsqrtps(v, T.rsqrt);
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

// v_cvt_i32_f32: truncation, saturating, NaN -> 0.  Equals cvttps2dq wherever the value is finite and inside the int
// range; the two also agree after a clamp to [0, n] for NaN and for values below -2^31 (INT_MIN and 0 / INT_MIN both
// clamp to 0).  They differ for +inf and values >= 2^31 (INT_MIN vs INT_MAX): callers state why those cannot occur.
__device__ __forceinline__ int32_t cvt_i32_sat(float f)
{
    int32_t r;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(f));
    return r;
}

// minps / maxps: second operand wins when unordered.
__device__ __forceinline__ float fmin_x86(float a, float b) { return (a < b) ? a : b; }
__device__ __forceinline__ float fmax_x86(float a, float b) { return (a > b) ? a : b; }
__device__ __forceinline__ float fclamp_x86(float v, float lo, float hi) { return fmin_x86(fmax_x86(v, lo), hi); }
// The same clamp for bounds that are ordinary numbers (lo < hi): maxps(v, lo) returns lo for a NaN v, exactly like
// v_max_f32 (maxNum), and is an ordinary maximum otherwise; its result is never NaN, so the minps that follows is an
// ordinary minimum too.  Two instructions instead of two compare / wait / select triples.  (Only the sign of a zero
// result can differ -- v = -0 against lo = +0 -- and every caller converts or squares the result.)
__device__ __forceinline__ float fclamp_num(float v, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(v, lo), hi); }

__device__ __forceinline__ int32_t iclamp(int32_t v, int32_t lo, int32_t hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ float sq(float v) { return v * v; }

} // namespace itw
