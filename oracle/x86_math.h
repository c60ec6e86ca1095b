/*
 * oracle/x86_math.h -- TEST INFRASTRUCTURE (parity oracle), not product code.
 *
 * The pinned arithmetic ("ISPC x86 model") that the reference's shipped build
 * computes with:  ispc -O2 --target=sse2,sse4,avx,avx2 --opt=fast-math
 * (/root/reference/IntelCompressionPlugin/IntelTextureWorks.vcxproj:388-393).
 *
 *  S2  fast-math front-end rewrite: binary  x / y  ->  x * rcp(y),
 *      x / const -> x * (1.f/const); compound  a /= b  stays an IEEE divide
 *      (kernel.ispc:1158 is the only such site).
 *  S3  rcp(v)   = r*(2 - v*r),            r  = rcpps(v)     (3 rounded ops)
 *      rsqrt(v) = 0.5*(is*(3 - (v*is)*is)), is = rsqrtps(v)  (5 rounded ops)
 *      seeds = Intel table functions, see tools/extract_x86_luts.c (verified
 *      against the instructions for all 2^32 inputs on an Intel Xeon).
 *  S4  no FMA contraction (sse2/sse4/avx targets): build with -ffp-contract=off.
 *  S5  (int)f = cvttps2dq: truncate; NaN / out of range -> INT_MIN.
 *  S6  min/max = minps/maxps: (a<b)?a:b / (a>b)?a:b  (returns b on NaN).
 *
 * PARITY STATUS: /root/reference holds no golden outputs and ispc is not installable
 * here, so this model (not an ISPC binary) defines "bit-exact" for the project.  The
 * ALGORITHM built on it is pinned: the reference's kernel.ispc, compiled as one scalar
 * program instance over these same functions (oracle/ref_build/ispc_as_cpp/ ->
 * oracle/_ref/libispc_texcomp_ref_full.so), emits the oracle's bytes
 * (tests/test_reference_kernel_source.py).  What stays an assumption is S2-S6 itself:
 * what the ispc compiler and its stdlib do.  See DESIGN.md section 5.
 */
#ifndef ORACLE_X86_MATH_H
#define ORACLE_X86_MATH_H

#include <stdint.h>
#include <string.h>
#include <math.h>
#include "x86_luts.h"

static inline uint32_t xm_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float    xm_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* rcpps: sign/exponent arithmetic + 2048-entry mantissa table */
static inline float x86_rcpps(float v)
{
    uint32_t x = xm_f2u(v);
    uint32_t s = x & 0x80000000u, e = (x >> 23) & 255u, m = x & 0x7fffffu;
    if (e == 255u) return xm_u2f(m ? (x | 0x00400000u) : s);
    if (e == 0u)   return xm_u2f(s | 0x7f800000u);
    uint32_t t = X86_RCP_SEED[m >> 12];
    int32_t re = (int32_t)((t >> 23) & 255u) + 127 - (int32_t)e;
    if (re <= 0) return xm_u2f(s);
    return xm_u2f(s | ((uint32_t)re << 23) | (t & 0x7fffffu));
}

/* rsqrtps: exponent parity + 1024-entry mantissa table */
static inline float x86_rsqrtps(float v)
{
    uint32_t x = xm_f2u(v);
    uint32_t s = x & 0x80000000u, e = (x >> 23) & 255u, m = x & 0x7fffffu;
    if (e == 255u && m) return xm_u2f(x | 0x00400000u);
    if (e == 0u)        return xm_u2f(s | 0x7f800000u);
    if (s)              return xm_u2f(0xffc00000u);
    if (e == 255u)      return 0.0f;
    uint32_t odd = (e & 1u) ? 0u : 1u;
    uint32_t t = X86_RSQRT_SEED[(odd << 10) | (m >> 13)];
    int32_t  k = ((int32_t)e - (int32_t)(127 + odd)) / 2;
    return xm_u2f(t - ((uint32_t)k << 23));
}

/*
 * ---- arithmetic-model switches: STUDY ONLY (`make -C oracle variants`) ---------------------------
 * The default build (no switch) is the pinned model every parity test uses.  The three assumptions nobody can verify
 * without an ispc binary each get a compile-time switch so that their weight can be MEASURED
 * (tools/arith_sensitivity.py -> profiles/arith_sensitivity.txt):
 *   ORACLE_MODEL_DIV1158_RCP  `proj /= div` (kernel.ispc:1158) lowered like every other division: proj * rcp(div)
 *   ORACLE_MODEL_IEEE         rcp(v) = 1.0f/v, rsqrt(v) = 1.0f/sqrtf(v) (correctly rounded, no table seeds, no Newton)
 *   -ffp-contract=fast -mfma  (a compiler flag, not a macro) the avx2 target's licence to fuse a*b+c
 *                             (IntelTextureWorks.vcxproj:388 builds sse2,sse4,avx,avx2); which sums gcc fuses need
 *                             not be the ones ispc/LLVM would fuse -- the variant measures sensitivity, not a target.
 */
#ifdef ORACLE_MODEL_DIV1158_RCP
#define ORACLE_DIV_1158(proj, div) ((proj) * ispc_rcp(div))
#else
#define ORACLE_DIV_1158(proj, div) ((proj) / (div))          /* true IEEE divide */
#endif

#ifdef ORACLE_MODEL_IEEE
static inline float ispc_rcp(float v)   { return 1.0f / v; }
static inline float ispc_rsqrt(float v) { return 1.0f / sqrtf(v); }
#else
/* ISPC stdlib rcp()/rsqrt() on the sse/avx targets: seed + one Newton step */
static inline float ispc_rcp(float v)
{
    float r = x86_rcpps(v);
    float t = v * r;
    t = 2.0f - t;
    return r * t;
}

static inline float ispc_rsqrt(float v)
{
    float is = x86_rsqrtps(v);
    float a = v * is;
    a = a * is;
    a = 3.0f - a;
    a = is * a;
    return 0.5f * a;
}
#endif

/* cvttss2si */
static inline int32_t f2i_x86(float f)
{
    if (!(f >= -2147483648.0f && f < 2147483648.0f)) return INT32_MIN; /* NaN fails both tests */
    return (int32_t)f;
}

/* minps / maxps operand order: result = second operand when unordered */
static inline float fmin_x86(float a, float b) { return (a < b) ? a : b; }
static inline float fmax_x86(float a, float b) { return (a > b) ? a : b; }
static inline float fclamp_x86(float v, float lo, float hi) { return fmin_x86(fmax_x86(v, lo), hi); }

static inline int32_t imin(int32_t a, int32_t b) { return (a < b) ? a : b; }
static inline int32_t imax(int32_t a, int32_t b) { return (a > b) ? a : b; }
static inline int32_t iclamp(int32_t v, int32_t lo, int32_t hi) { return imin(imax(v, lo), hi); }

static inline float sqf(float v) { return v * v; }

#endif
