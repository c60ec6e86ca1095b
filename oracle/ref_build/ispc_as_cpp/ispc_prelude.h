// TEST INFRASTRUCTURE (oracle/_ref build only; the product never includes, links or calls this).
//
// kernel.ispc is ISPC source and the ispc compiler is not available in this image.  Its language subset, however, is
// C with five extras (uniform / varying qualifiers, `foreach`, `export`, sized int type names, float literals without a
// dot), and a gang's program instances never interact on this path, so ONE instance is an ordinary scalar
// program.  translate.py rewrites the extras textually at build time (the result lives only in oracle/_ref/), and
// this header supplies what the ispc compiler and its standard library would: the meaning of `float` arithmetic under
// `--opt=fast-math` and the handful of stdlib functions the file calls.  What the result PINS: the oracle's reading of
// kernel.ispc, line by line, by the reference's own source.  What it cannot pin: the compiler / stdlib semantics
// themselves, which are the same pinned model as everywhere else (oracle/x86_math.h).
//
//   S1  float literals are fp32                     translate.py suffixes them; a double reaching ispc_float is a compile error
//   S2  x / y -> x * rcp(y);  x / const -> x * (1/const);  the compound  a /= b  stays an IEEE divide
//       (-DISPC_DIVASSIGN_RCP flips the last one)
//   S3  rcp / rsqrt = RCPPS / RSQRTPS seed + one Newton step (oracle/x86_math.h; -DORACLE_MODEL_IEEE for the IEEE model)
//   S4  no FMA                                      the Makefile passes -ffp-contract=off
//   S5  (int)f = cvttps2dq
//   S6  min / max = minps / maxps operand order
//   S7  integers wrap                               the Makefile passes -fwrapv
#ifndef ISPC_PRELUDE_H
#define ISPC_PRELUDE_H
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <stdio.h>
extern "C" {
#include "x86_math.h"
}

#define ISPC_INLINE inline __attribute__((always_inline))

typedef float ispc_raw;       // the builtin type, still nameable after `#define float` below
struct ispc_float {
    float v;
    ispc_float() = default;
    ISPC_INLINE ispc_float(float x) : v(x) {}
    ISPC_INLINE ispc_float(int x) : v((float)x) {}
    ISPC_INLINE ispc_float(unsigned int x) : v((float)x) {}
    ispc_float(double) = delete;                                   // an unsuffixed literal slipped through translate.py
    ISPC_INLINE operator int() const { return f2i_x86(v); }        // S5; implicit like in ISPC (`int best_err = err1;`, kernel.ispc:1178)
};
static_assert(sizeof(ispc_float) == 4, "ispc_float is a plain fp32");

// Arithmetic.  ISPC promotes int to float in mixed expressions; C++ would find `F op int` ambiguous between that and
// the implicit F -> int conversion, so every mixed pair gets its exact overload.
#define ISPC_MIXED(RET, NAME, EXPR)                                                                                       \
    ISPC_INLINE RET NAME(ispc_float a, ispc_float b) { return EXPR; }                                                      \
    ISPC_INLINE RET NAME(ispc_float a, int b_) { const ispc_float b(b_); return EXPR; }                                    \
    ISPC_INLINE RET NAME(int a_, ispc_float b) { const ispc_float a(a_); return EXPR; }                                    \
    ISPC_INLINE RET NAME(ispc_float a, unsigned int b_) { const ispc_float b(b_); return EXPR; }                           \
    ISPC_INLINE RET NAME(unsigned int a_, ispc_float b) { const ispc_float a(a_); return EXPR; }                           \
    ISPC_INLINE RET NAME(ispc_float a, ispc_raw b_) { const ispc_float b(b_); return EXPR; }                                  \
    ISPC_INLINE RET NAME(ispc_raw a_, ispc_float b) { const ispc_float a(a_); return EXPR; }
ISPC_MIXED(ispc_float, operator+, ispc_float(a.v + b.v))
ISPC_MIXED(ispc_float, operator-, ispc_float(a.v - b.v))
ISPC_MIXED(ispc_float, operator*, ispc_float(a.v * b.v))
ISPC_MIXED(bool, operator<, a.v < b.v)
ISPC_MIXED(bool, operator>, a.v > b.v)
ISPC_MIXED(bool, operator<=, a.v <= b.v)
ISPC_MIXED(bool, operator>=, a.v >= b.v)
ISPC_MIXED(bool, operator==, a.v == b.v)
ISPC_MIXED(bool, operator!=, a.v != b.v)
ISPC_INLINE ispc_float operator-(ispc_float a) { return xm_u2f(xm_f2u(a.v) ^ 0x80000000u); }
// S2: binary division under --opt=fast-math.  A divisor that is a compile-time constant becomes a multiplication by
// its IEEE reciprocal; anything else goes through rcp().  The int / float overloads exist for literal divisors
// (`x / 16`, `x / 255f`); should a variable arrive there, __builtin_constant_p sends it to rcp() like ispc would.
ISPC_INLINE ispc_float operator/(ispc_float a, ispc_float b) { return a.v * ispc_rcp(b.v); }
ISPC_INLINE ispc_float operator/(ispc_float a, float c) { return __builtin_constant_p(c) ? a.v * (1.0f / c) : a.v * ispc_rcp(c); }
ISPC_INLINE ispc_float operator/(ispc_float a, int c) { return __builtin_constant_p(c) ? a.v * (1.0f / (float)c) : a.v * ispc_rcp((float)c); }
ISPC_INLINE ispc_float operator/(ispc_float a, unsigned int c) { return __builtin_constant_p(c) ? a.v * (1.0f / (float)c) : a.v * ispc_rcp((float)c); }
ISPC_INLINE ispc_float operator/(float a, ispc_float b) { return a * ispc_rcp(b.v); }
ISPC_INLINE ispc_float operator/(int a, ispc_float b) { return (float)a * ispc_rcp(b.v); }
ISPC_INLINE ispc_float operator/(unsigned int a, ispc_float b) { return (float)a * ispc_rcp(b.v); }
#define ISPC_COMPOUND(OP, EXPR)                                                                                            \
    ISPC_INLINE ispc_float& operator OP(ispc_float& a, ispc_float b) { a.v = EXPR; return a; }                             \
    ISPC_INLINE ispc_float& operator OP(ispc_float& a, int b_) { const ispc_float b(b_); a.v = EXPR; return a; }           \
    ISPC_INLINE ispc_float& operator OP(ispc_float& a, unsigned int b_) { const ispc_float b(b_); a.v = EXPR; return a; }  \
    ISPC_INLINE ispc_float& operator OP(ispc_float& a, float b_) { const ispc_float b(b_); a.v = EXPR; return a; }
ISPC_COMPOUND(+=, a.v + b.v)
ISPC_COMPOUND(-=, a.v - b.v)
ISPC_COMPOUND(*=, a.v * b.v)
#ifdef ISPC_DIVASSIGN_RCP
ISPC_COMPOUND(/=, a.v * ispc_rcp(b.v))
#else
ISPC_COMPOUND(/=, a.v / b.v)                 // the compound form is not rewritten by fast-math (S2): IEEE divide
#endif

#define assert(x) ((void)0)          /* ispc drops assert() in release builds as well */

// the language extras that are plain tokens
#define uniform
#define varying
#define export extern "C"
#define int8 char
#define int16 short
#define int32 int
#define int64 long long
#define float ispc_float
#endif
