/*
 * oracle/ref_build/dxmath_stub/directxmath.h -- TEST INFRASTRUCTURE.
 *
 * The ~30 DirectXMath entry points the reference's block codecs (3rdParty/DirectXTex/DirectXTex/BC.cpp, BC4BC5.cpp,
 * BC6HBC7.cpp, BC.h) touch, restated in scalar C++ after the library's documented no-intrinsics semantics, so that those
 * files compile UNMODIFIED on Linux (the Windows SDK's DirectXMath is not vendored in the reference tree).  Used to build
 * oracle/_ref/libdxtex_bc_ref.so: the reference's own decoders (D3DXDecodeBC1/3/4U/5U/6HU/7) and its BC4/BC5 encoders, as
 * pins for the from-spec decoders and the BC4/BC5 restatement in oracle/.  Everything here is glue (loads, stores, lane
 * arithmetic); the block-level algorithms are the reference's.
 */
#pragma once
#include <stdint.h>
#include <string.h>
#include <math.h>

#define DIRECTX_CTOR_DEFAULT =default;
#define XM_CALLCONV

namespace DirectX {

struct XMVECTOR {
    union { float f[4]; uint32_t u[4]; int32_t i[4]; };
};
typedef const XMVECTOR FXMVECTOR;

struct XMFLOAT4 { float x, y, z, w; XMFLOAT4() = default; XMFLOAT4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {} };
struct alignas(16) XMFLOAT4A : public XMFLOAT4 { XMFLOAT4A() = default; XMFLOAT4A(float a, float b, float c, float d) : XMFLOAT4(a, b, c, d) {} };
struct XMINT4 { int32_t x, y, z, w; };
struct XMVECTORF32 { union { float f[4]; XMVECTOR v; }; operator XMVECTOR() const { return v; } };
struct XMVECTORU32 { union { uint32_t u[4]; XMVECTOR v; }; operator XMVECTOR() const { return v; } };

static const XMVECTORF32 g_XMIdentityR3 = {{{0.f, 0.f, 0.f, 1.f}}};
static const XMVECTORU32 g_XMSelect1110 = {{{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u}}};

inline XMVECTOR XMVectorSet(float x, float y, float z, float w) { XMVECTOR v; v.f[0] = x; v.f[1] = y; v.f[2] = z; v.f[3] = w; return v; }
inline XMVECTOR XMVectorZero() { return XMVectorSet(0.f, 0.f, 0.f, 0.f); }
inline float XMVectorGetX(FXMVECTOR v) { return v.f[0]; }
inline XMVECTOR XMVectorSetW(FXMVECTOR v, float w) { XMVECTOR r = v; r.f[3] = w; return r; }
inline XMVECTOR XMVectorSubtract(FXMVECTOR a, FXMVECTOR b) { XMVECTOR r; for (int i = 0; i < 4; i++) r.f[i] = a.f[i] - b.f[i]; return r; }
inline XMVECTOR XMVectorMultiply(FXMVECTOR a, FXMVECTOR b) { XMVECTOR r; for (int i = 0; i < 4; i++) r.f[i] = a.f[i] * b.f[i]; return r; }
/* V0 + t * (V1 - V0), as XMVectorMultiplyAdd(Length, Scale, V0): a multiply then an add */
inline XMVECTOR XMVectorLerp(FXMVECTOR a, FXMVECTOR b, float t) { XMVECTOR r; for (int i = 0; i < 4; i++) { float l = b.f[i] - a.f[i]; l = l * t; r.f[i] = l + a.f[i]; } return r; }
inline XMVECTOR XMVector3Dot(FXMVECTOR a, FXMVECTOR b) { const float d = a.f[0] * b.f[0] + a.f[1] * b.f[1] + a.f[2] * b.f[2]; return XMVectorSet(d, d, d, d); }
inline XMVECTOR XMVector4Dot(FXMVECTOR a, FXMVECTOR b) { const float d = a.f[0] * b.f[0] + a.f[1] * b.f[1] + a.f[2] * b.f[2] + a.f[3] * b.f[3]; return XMVectorSet(d, d, d, d); }
template <uint32_t X, uint32_t Y, uint32_t Z, uint32_t W>
inline XMVECTOR XMVectorSwizzle(FXMVECTOR v) { XMVECTOR r; r.u[0] = v.u[X]; r.u[1] = v.u[Y]; r.u[2] = v.u[Z]; r.u[3] = v.u[W]; return r; }
inline XMVECTOR XMVectorSelect(FXMVECTOR a, FXMVECTOR b, FXMVECTOR c) { XMVECTOR r; for (int i = 0; i < 4; i++) r.u[i] = (a.u[i] & ~c.u[i]) | (b.u[i] & c.u[i]); return r; }

inline XMVECTOR XMLoadFloat4(const XMFLOAT4* p) { return XMVectorSet(p->x, p->y, p->z, p->w); }
inline void XMStoreFloat4(XMFLOAT4* p, FXMVECTOR v) { p->x = v.f[0]; p->y = v.f[1]; p->z = v.f[2]; p->w = v.f[3]; }
inline void XMStoreFloat4A(XMFLOAT4A* p, FXMVECTOR v) { XMStoreFloat4(p, v); }
inline XMVECTOR XMLoadSInt4(const XMINT4* p) { XMVECTOR v; v.i[0] = p->x; v.i[1] = p->y; v.i[2] = p->z; v.i[3] = p->w; return v; }   /* raw integers */

} // namespace DirectX
