/*
 * itw_bc45.h -- BC4_UNORM / BC5_UNORM block encoding on the GPU.
 *
 * The two formats the plugin offers that bypass ispc_texcomp: IntelPlugin.cpp:120-141 converts the document to an
 * RGBA8 scratch image (BC4: the red plane in every colour channel, BC5: red and green) and IntelPlugin.cpp:271-273
 * calls DirectXTex
 *     Compress(images, nimages, metadata, DXGI_FORMAT_BC4_UNORM | DXGI_FORMAT_BC5_UNORM, TEX_COMPRESS_DEFAULT, 0.5f, out)
 * (3rdParty/DirectXTex/DirectXTex/DirectXTexCompress.cpp:607, :73-186), which encodes each 4x4 block with
 * D3DXEncodeBC4U / D3DXEncodeBC5U (3rdParty/DirectXTex/DirectXTex/BC4BC5.cpp:403, :481).  The entry points below replace
 * that per-image call with the calling convention of the library's other encoders, so the trampolines of
 * itw_dispatch.h and a patched plugin can treat all six formats alike.
 *
 * src     RGBA8 surface (ispc_texcomp.h rgba_surface), R in the lowest byte.  BC4 encodes R; BC5 encodes R then G.
 *         Any width, height >= 1: like DirectXTex (and unlike the ISPC formats) partial blocks are kept, their missing
 *         columns / rows filled from source column / row {0,0,0,1}[i] (DirectXTexCompress.cpp:140-168).
 * dst     ceil(width/4) * ceil(height/4) blocks in raster order, 8 bytes (BC4) or 16 bytes (BC5: R block, G block),
 *         tightly packed (DirectXTex's pitch rule, DirectXTexUtil.cpp:601-619).
 * Host or device pointers, threading, streams and error behaviour: exactly as CompressBlocksBC1 (ispc_texcomp.h).
 */
#ifndef ITW_BC45_H
#define ITW_BC45_H

#include "ispc_texcomp.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only what the headers declare is exported */

void CompressBlocksBC4(const rgba_surface* src, uint8_t* dst);
void CompressBlocksBC5(const rgba_surface* src, uint8_t* dst);

/* The first BC4 / BC5 call on a device builds a 1 MiB index table there (csrc/bc4_bc5.hip): that one call allocates device memory and is not
 * stream-capturable; itwWarmupBC45() does the same ahead of time (see itw_amd.h for the asynchronous contract of device-pointer calls). */
void itwWarmupBC45(void);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
