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

void CompressBlocksBC4(const rgba_surface* src, uint8_t* dst);
void CompressBlocksBC5(const rgba_surface* src, uint8_t* dst);

/* Test hook (tests/test_gpu_bc45_index_table.py).  The encoders evaluate FindClosestUNORM (BC4BC5.cpp:314-337) through a table
 * the current device builds once by running that search, as written, for all 65 536 endpoint pairs: per pair the 7 texel
 * codes where the chosen index changes and the 8 indices of the runs in between (csrc/bc4_bc5.hip).  Copies the table --
 * 65 536 entries x 4 words {256 - start of run 1..4, one byte each; the same for runs 5..7 with the number of runs in
 * the top byte; indices of runs 0..3, one byte each; indices of runs 4..7} -- to host memory.  Returns 0, or -1 on failure (error mode "return"). */
int itwTestBc45IndexTable(uint32_t* host_out);

#ifdef __cplusplus
}
#endif
#endif
