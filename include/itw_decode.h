/*
 * itw_decode.h -- BC1 / BC3 / BC4 / BC5 / BC7 / BC6H(unsigned) block decoding on the GPU: the step immediately after the ABI in
 * the reference's preview and load paths, where DirectXTex's Decompress() (D3DXDecodeBC1/BC3/BC7/BC6HU,
 * 3rdParty/DirectXTex/DirectXTex/BC.cpp, BC6HBC7.cpp:1077-1210, 1937-2140) turns the blocks back into texels
 * (IntelPlugin.cpp:1059, 2558).  Written from the format definitions; used here for preview-style round trips and
 * for whole-surface validity / PSNR checks of the encoder's output without leaving HBM.
 */
#ifndef ITW_DECODE_H
#define ITW_DECODE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only what the headers declare is exported */

/* Decodes (width/4)*(height/4) tightly packed blocks in raster block order into a surface of `out_stride` bytes per
 * texel row: RGBA8 for BC1 / BC3 / BC7, (R,0,0,255) / (R,G,0,255) RGBA8 for BC4 / BC5 (the layout D3DXDecodeBC4U/BC5U
 * produce, BC4BC5.cpp:373-385, 449-462; 8-bit values by the format's integer definition, rounded to nearest),
 * RGBA16F bit patterns for BC6H (alpha = 1.0 = 0x3C00).  width and height are multiples of 4 (a BC4/BC5 stream of a
 * partial surface decodes to the padded size).
 * dxgi_format: one of the ITW_DXGI_FORMAT_BC* values of itw_dispatch.h (71,72,77,78,80,83,95,96,98,99).
 * `blocks`, `out`, `modes` are host or device pointers (host pointers are staged, the call then returns synchronised;
 * all-device calls are asynchronous on the calling thread's stream, itwSetStream).
 * `modes` (optional, may be NULL): one int32 per block -- BC7: mode 0..7, -1 for the reserved all-zero-prefix block;
 * BC6H: mode 0..13 in kernel.ispc's numbering, -1 for a reserved prefix; BC1/BC3: 0.
 * Width and height must be multiples of 4, except for BC4 / BC5, whose streams may end in partial blocks (cropped on store).
 * Returns 0, or -1 for an unsupported format / misaligned sizes. */
int itwDecodeBlocks(int dxgi_format, const uint8_t* blocks, int width, int height, uint8_t* out, int64_t out_stride, int32_t* modes);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
