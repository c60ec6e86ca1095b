/*
 * oracle/ref_build/dxtex_glue.cpp -- TEST INFRASTRUCTURE.
 *
 * C face of oracle/_ref/libdxtex_bc_ref.so = the reference's own block codecs, compiled unmodified from
 * /root/reference/3rdParty/DirectXTex/DirectXTex/{BC.cpp, BC4BC5.cpp, BC6HBC7.cpp} (with dxmath_stub/ standing in for the
 * Windows SDK's DirectXMath).  Used by tests/test_reference_codecs.py:
 *   * D3DXDecodeBC1 / BC3 / BC4U / BC5U / BC6HU / BC7 (BC.cpp:327-, BC4BC5.cpp:373-, BC6HBC7.cpp:2790-) decode the streams
 *     this project's encoders emit -- the reference's reading of every block, against the from-spec decoders in oracle/ and
 *     csrc/decode.hip;
 *   * D3DXEncodeBC4U / BC5U (BC4BC5.cpp:403, 481) -- the encoder the plugin itself calls for these two formats
 *     (IntelPlugin.cpp:271-273) -- against oracle/bc4_bc5.c and csrc/bc4_bc5.hip.
 */
#include "directxtexp.h"
#include "BC.h"

using namespace DirectX;

extern "C" {

/* kind: 1 BC1, 3 BC3, 4 BC4U, 5 BC5U, 6 BC6HU, 7 BC7.  out: 16 texels x RGBA floats, as the reference returns them. */
int dxtex_ref_decode(int kind, const uint8_t* block, float* out)
{
    XMVECTOR px[16];
    switch (kind) {
    case 1: D3DXDecodeBC1(px, block); break;
    case 3: D3DXDecodeBC3(px, block); break;
    case 4: D3DXDecodeBC4U(px, block); break;
    case 5: D3DXDecodeBC5U(px, block); break;
    case 6: D3DXDecodeBC6HU(px, block); break;
    case 7: D3DXDecodeBC7(px, block); break;
    default: return -1;
    }
    for (int i = 0; i < 16; i++) for (int c = 0; c < 4; c++) out[4 * i + c] = px[i].f[c];
    return 0;
}

/* texels: 16 x (r, g) floats in [0, 1] exactly as DirectXTex's loader hands them over; nch = 1 (BC4U) or 2 (BC5U) */
void dxtex_ref_encode_bc45(int nch, const float* rg, uint8_t* out)
{
    XMVECTOR px[16];
    for (int i = 0; i < 16; i++) px[i] = XMVectorSet(rg[2 * i], rg[2 * i + 1], 0.f, 1.f);
    if (nch == 1) D3DXEncodeBC4U(out, px, 0); else D3DXEncodeBC5U(out, px, 0);
}

}
