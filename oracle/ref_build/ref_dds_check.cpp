/*
 * oracle/ref_build/ref_dds_check.cpp -- TEST INFRASTRUCTURE.
 *
 * Reads a DDS file header through the REFERENCE's own container definitions -- DirectXTex/DDS.h compiled unmodified from
 * /root/reference (struct DDS_HEADER, DDS_HEADER_DXT10, the DDSPF_* pixel formats and flag macros) -- and checks the file
 * written by this project's container code (csrc/dds.hip, include/itw_dds.h) against the rules DirectXTex's writer
 * applies to block-compressed textures (DirectXTexDDS.cpp:441-675: flags, caps, linear size; restated as checks below).
 *
 *   ref_dds_check <file.dds> <expect: DXT1|DXT5|BC4U|BC5U|DX10> <dxgi> <width> <height> <mips> <cube 0|1> <array>
 * prints OK or the first field that differs; exit code 0 / 1.
 */
#define __declspec(x)
#include "DDS.h"                 /* the reference's header */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace DirectX;

#define CHECK(cond, ...) do { if (!(cond)) { printf("MISMATCH " __VA_ARGS__); printf("\n"); return 1; } } while (0)

int main(int argc, char** argv)
{
    if (argc != 9) { fprintf(stderr, "usage: %s file expect dxgi w h mips cube array\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    std::vector<unsigned char> buf(4 + sizeof(DDS_HEADER) + sizeof(DDS_HEADER_DXT10));
    const size_t got = fread(buf.data(), 1, buf.size(), f);
    fclose(f);
    const char* expect = argv[2];
    const uint32_t dxgi = (uint32_t)atoi(argv[3]), w = (uint32_t)atoi(argv[4]), h = (uint32_t)atoi(argv[5]), mips = (uint32_t)atoi(argv[6]);
    const bool cube = atoi(argv[7]) != 0;
    const uint32_t array = (uint32_t)atoi(argv[8]);
    CHECK(got >= 4 + sizeof(DDS_HEADER), "file shorter than magic + DDS_HEADER");
    uint32_t magic;
    memcpy(&magic, buf.data(), 4);
    CHECK(magic == DDS_MAGIC, "magic %08x", magic);
    DDS_HEADER hd;
    memcpy(&hd, buf.data() + 4, sizeof hd);
    CHECK(hd.dwSize == sizeof(DDS_HEADER), "dwSize %u", hd.dwSize);
    CHECK(hd.ddspf.dwSize == sizeof(DDS_PIXELFORMAT), "ddspf.dwSize %u", hd.ddspf.dwSize);
    CHECK(hd.dwWidth == w && hd.dwHeight == h, "size %ux%u", hd.dwWidth, hd.dwHeight);
    const DDS_PIXELFORMAT* pf = !strcmp(expect, "DXT1") ? &DDSPF_DXT1 : !strcmp(expect, "DXT5") ? &DDSPF_DXT5 :
                                !strcmp(expect, "BC4U") ? &DDSPF_BC4_UNORM : !strcmp(expect, "BC5U") ? &DDSPF_BC5_UNORM : &DDSPF_DX10;
    CHECK(!memcmp(&hd.ddspf, pf, sizeof(DDS_PIXELFORMAT)), "ddspf differs from the reference's DDSPF_%s", expect);
    /* block-compressed: linear size of the top level, DirectXTexDDS.cpp:563-570 + DirectXTexUtil.cpp:601-619 */
    uint32_t flags = DDS_HEADER_FLAGS_TEXTURE | DDS_HEADER_FLAGS_LINEARSIZE;
    uint32_t caps = DDS_SURFACE_FLAGS_TEXTURE;
    if (mips > 0) flags |= DDS_HEADER_FLAGS_MIPMAP;                   /* DirectXTexDDS.cpp:540-542: always, mipLevels >= 1 */
    if (mips > 1) caps |= DDS_SURFACE_FLAGS_MIPMAP;                   /* :551-552 */
    if (cube) caps |= DDS_SURFACE_FLAGS_CUBEMAP;                      /* :578-582 */
    CHECK(hd.dwFlags == flags, "dwFlags %08x want %08x", hd.dwFlags, flags);
    CHECK(hd.dwCaps == caps, "dwCaps %08x want %08x", hd.dwCaps, caps);
    CHECK(hd.dwCaps2 == (cube ? (uint32_t)DDS_CUBEMAP_ALLFACES : 0u), "dwCaps2 %08x", hd.dwCaps2);
    CHECK(hd.dwMipMapCount == mips, "dwMipMapCount %u", hd.dwMipMapCount);
    CHECK(hd.dwDepth == 1, "dwDepth %u", hd.dwDepth);
    const uint32_t bpb = (dxgi == 71 || dxgi == 72 || dxgi == 80) ? 8 : 16;
    const uint32_t linear = ((w + 3) / 4 > 0 ? (w + 3) / 4 : 1) * ((h + 3) / 4 > 0 ? (h + 3) / 4 : 1) * bpb;
    CHECK(hd.dwPitchOrLinearSize == linear, "dwPitchOrLinearSize %u want %u", hd.dwPitchOrLinearSize, linear);
    if (pf == &DDSPF_DX10) {
        CHECK(got >= 4 + sizeof(DDS_HEADER) + sizeof(DDS_HEADER_DXT10), "DX10 extension missing");
        DDS_HEADER_DXT10 ex;
        memcpy(&ex, buf.data() + 4 + sizeof(DDS_HEADER), sizeof ex);
        CHECK((uint32_t)ex.dxgiFormat == dxgi, "dxgiFormat %u", (uint32_t)ex.dxgiFormat);
        CHECK(ex.resourceDimension == DDS_DIMENSION_TEXTURE2D, "resourceDimension %u", ex.resourceDimension);
        CHECK(ex.miscFlag == (cube ? (uint32_t)DDS_RESOURCE_MISC_TEXTURECUBE : 0u), "miscFlag %u", ex.miscFlag);
        CHECK(ex.arraySize == array, "arraySize %u want %u", ex.arraySize, array);
    }
    printf("OK\n");
    return 0;
}
