/* TEST INFRASTRUCTURE: DirectX::PackedVector subset for the reference's block codecs (see directxmath.h here). */
#pragma once
#include "directxmath.h"

namespace DirectX { namespace PackedVector {

typedef uint16_t HALF;
struct XMHALF4 { HALF x, y, z, w; };
struct XMUBYTE4 { uint8_t x, y, z, w; };
struct XMU565 { union { struct { uint16_t x : 5; uint16_t y : 6; uint16_t z : 5; }; uint16_t v; }; };

/* IEEE half <-> float, DirectXMath's scalar routines (round to nearest even on the way down) */
inline float XMConvertHalfToFloat(HALF h)
{
    uint32_t mant = h & 0x03FFu, exp = h & 0x7C00u;
    if (exp == 0x7C00u) exp = 0x8Fu;                                   /* inf / NaN */
    else if (exp != 0) exp = (h >> 10) & 0x1Fu;                        /* normal */
    else if (mant != 0) {                                              /* denormal: normalise */
        exp = 1;
        do { exp--; mant <<= 1; } while ((mant & 0x0400u) == 0);
        mant &= 0x03FFu;
    } else exp = (uint32_t)-112;                                       /* zero */
    const uint32_t r = ((h & 0x8000u) << 16) | ((exp + 112u) << 23) | (mant << 13);
    float f; memcpy(&f, &r, 4); return f;
}
inline HALF XMConvertFloatToHalf(float f)
{
    uint32_t i; memcpy(&i, &f, 4);
    const uint32_t sign = (i & 0x80000000u) >> 16;
    i &= 0x7FFFFFFFu;
    uint32_t r;
    if (i > 0x477FE000u) {                                             /* too large: inf / NaN */
        if (((i & 0x7F800000u) == 0x7F800000u) && ((i & 0x7FFFFFu) != 0)) r = 0x7FFFu; else r = 0x7C00u;
    } else if (!i) r = 0;
    else {
        if (i < 0x38800000u) {                                         /* becomes a denormal half */
            const uint32_t shift = 113u - (i >> 23);
            i = shift < 24 ? (0x800000u | (i & 0x7FFFFFu)) >> shift : 0;
        } else i += 0xC8000000u;                                       /* rebias */
        r = ((i + 0x0FFFu + ((i >> 13) & 1u)) >> 13) & 0x7FFFu;
    }
    return (HALF)(r | sign);
}
inline void XMStoreHalf4(XMHALF4* p, FXMVECTOR v) { p->x = XMConvertFloatToHalf(v.f[0]); p->y = XMConvertFloatToHalf(v.f[1]); p->z = XMConvertFloatToHalf(v.f[2]); p->w = XMConvertFloatToHalf(v.f[3]); }
inline XMVECTOR XMLoadUByte4(const XMUBYTE4* p) { return XMVectorSet((float)p->x, (float)p->y, (float)p->z, (float)p->w); }
inline XMVECTOR XMLoadU565(const XMU565* p) { return XMVectorSet((float)(p->v & 0x1F), (float)((p->v >> 5) & 0x3F), (float)((p->v >> 11) & 0x1F), 0.f); }

}} // namespace
