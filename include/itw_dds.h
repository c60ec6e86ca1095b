/*
 * itw_dds.h -- DDS container for the encoder's block streams: the header DirectXTex's _EncodeDDSHeader writes for
 * the formats the plugin saves (3rdParty/DirectXTex/DirectXTex/DDS.h:38-236, DirectXTexDDS.cpp:441-675, data
 * order :1611-1700, pitch rule DirectXTexUtil.cpp:601-619), restated portably.  BC1_UNORM / BC3_UNORM use the legacy
 * 'DXT1' / 'DXT5' FourCC header (128 bytes incl. magic); the _SRGB variants, BC7 and BC6H carry the 'DX10' extension
 * (148 bytes).  Mip chains: top level first; cube maps: 6 faces, each with its full chain.
 */
#ifndef ITW_DDS_H
#define ITW_DDS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only what the headers declare is exported */

typedef struct ItwDdsDesc {
    uint32_t width, height;     /* top mip, texels */
    uint32_t mip_levels;        /* >= 1 */
    uint32_t dxgi_format;       /* DXGI_FORMAT_BCn_* value (71,72,77,78,80,83,95,96,98,99) */
    uint32_t is_cubemap;        /* 0 / 1: six faces */
    uint32_t array_size;        /* number of textures (cubes when is_cubemap); >= 1 */
} ItwDdsDesc;

/* bytes of one mip level: max(1,(w+3)/4) * max(1,(h+3)/4) * bytes-per-block */
size_t itwDdsLevelBytes(uint32_t dxgi_format, uint32_t width, uint32_t height);
/* header bytes (incl. the 4-byte magic): 128 or 148; 0 if the format is not one of the BCn formats above */
size_t itwDdsHeaderBytes(const ItwDdsDesc* desc);
/* header + all faces / levels */
size_t itwDdsFileBytes(const ItwDdsDesc* desc);
/* Writes the header into dst (capacity >= itwDdsHeaderBytes); returns bytes written, 0 on error. */
size_t itwDdsWriteHeader(const ItwDdsDesc* desc, uint8_t* dst, size_t capacity);
/* Parses a header; returns its size (offset of the first block), 0 if not a BCn DDS this library writes. */
size_t itwDdsReadHeader(const uint8_t* src, size_t size, ItwDdsDesc* desc);
/* Whole file: `levels[i]` points at the blocks of image i in file order (array item major, then face, then mip).
 * Returns bytes written (== itwDdsFileBytes) or 0. */
size_t itwDdsWriteFile(const ItwDdsDesc* desc, const uint8_t* const* levels, size_t nlevels, uint8_t* dst, size_t capacity);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
