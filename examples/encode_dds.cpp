// encode_dds -- a C++ host above the C ABI, the way the plugin's save path uses it (IntelPlugin.cpp:186-261, 816-884):
// raw texels -> pad to multiples of 4 -> CompressImageMT through the slice loop -> .DDS file.  Only include/*.h is used;
// the program links libispc_texcomp.so like the plugin links ispc_texcomp.lib.
//
//   encode_dds <format> <width> <height> <in.raw> <out.dds> [slice_pixels]
//     format : bc1 | bc3 | bc4 | bc5 | bc7_<profile> | bc6h_<profile>      (profiles: the GetProfile_* names)
//     in.raw : width*height tightly packed RGBA8 texels (RGBA16F bit patterns for bc6h_*)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "../include/ispc_texcomp.h"
#include "../include/itw_bc45.h"
#include "../include/itw_dds.h"
#include "../include/itw_dispatch.h"

namespace {

struct Format { const char* name; CompressionFunc* fn; int dxgi; int texel_bytes; bool pad; };

const Format kFormats[] = {
    {"bc1", CompressImageBC1, ITW_DXGI_FORMAT_BC1_UNORM, 4, true},
    {"bc3", CompressImageBC3, ITW_DXGI_FORMAT_BC3_UNORM, 4, true},
    {"bc4", CompressImageBC4, ITW_DXGI_FORMAT_BC4_UNORM, 4, false},      // DirectXTex formats keep partial blocks
    {"bc5", CompressImageBC5, ITW_DXGI_FORMAT_BC5_UNORM, 4, false},
    {"bc7_ultrafast", CompressImageBC7_ultrafast, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_veryfast", CompressImageBC7_veryfast, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_fast", CompressImageBC7_fast, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_basic", CompressImageBC7_basic, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_slow", CompressImageBC7_slow, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_alpha_ultrafast", CompressImageBC7_alpha_ultrafast, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_alpha_veryfast", CompressImageBC7_alpha_veryfast, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_alpha_fast", CompressImageBC7_alpha_fast, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_alpha_basic", CompressImageBC7_alpha_basic, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc7_alpha_slow", CompressImageBC7_alpha_slow, ITW_DXGI_FORMAT_BC7_UNORM, 4, true},
    {"bc6h_veryfast", CompressImageBC6H_veryfast, ITW_DXGI_FORMAT_BC6H_UF16, 8, true},
    {"bc6h_fast", CompressImageBC6H_fast, ITW_DXGI_FORMAT_BC6H_UF16, 8, true},
    {"bc6h_basic", CompressImageBC6H_basic, ITW_DXGI_FORMAT_BC6H_UF16, 8, true},
    {"bc6h_slow", CompressImageBC6H_slow, ITW_DXGI_FORMAT_BC6H_UF16, 8, true},
    {"bc6h_veryslow", CompressImageBC6H_veryslow, ITW_DXGI_FORMAT_BC6H_UF16, 8, true},
};

bool on_progress(int done, int total, void*)
{
    std::fprintf(stderr, "\rslice %d / %d", done, total);
    return true;                                                   // false would abort like the plugin's cancel button
}

} // namespace

int main(int argc, char** argv)
{
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s <format> <width> <height> <in.raw> <out.dds> [slice_pixels]\n", argv[0]);
        return 2;
    }
    const Format* f = nullptr;
    for (const Format& k : kFormats) if (std::strcmp(k.name, argv[1]) == 0) f = &k;
    const int width = std::atoi(argv[2]), height = std::atoi(argv[3]);
    if (!f || width < 1 || height < 1) { std::fprintf(stderr, "unknown format or bad size\n"); return 2; }
    const long long slice_pixels = argc > 6 ? std::atoll(argv[6]) : 0;

    std::vector<uint8_t> texels((size_t)width * height * f->texel_bytes);
    FILE* in = std::fopen(argv[4], "rb");
    if (!in || std::fread(texels.data(), 1, texels.size(), in) != texels.size()) { std::fprintf(stderr, "cannot read %s\n", argv[4]); return 1; }
    std::fclose(in);

    rgba_surface source = { texels.data(), width, height, width * f->texel_bytes };
    rgba_surface padded = source;
    if (f->pad && ((width | height) & 3)) padded = itwPadToMultipleOf4(&source, f->texel_bytes);     // IntelPlugin.cpp:893-928

    ItwDdsDesc desc = { (uint32_t)padded.width, (uint32_t)padded.height, 1, (uint32_t)f->dxgi, 0, 1 };
    std::vector<uint8_t> blocks(itwDdsLevelBytes(desc.dxgi_format, desc.width, desc.height));
    const int64_t pitch = (int64_t)((padded.width + 3) / 4) * GetBytesPerBlock(f->dxgi);
    const bool ok = itwCompressImageSliced(&padded, blocks.data(), pitch, f->fn, f->dxgi, /*multithreaded*/ true,
                                           slice_pixels, slice_pixels ? on_progress : nullptr, nullptr);
    if (padded.ptr != source.ptr) itwFreeSurface(&padded);
    if (!ok) { std::fprintf(stderr, "\ncompression aborted\n"); return 1; }

    std::vector<uint8_t> file(itwDdsFileBytes(&desc));
    const uint8_t* levels[1] = { blocks.data() };
    if (itwDdsWriteFile(&desc, levels, 1, file.data(), file.size()) != file.size()) { std::fprintf(stderr, "DDS assembly failed\n"); return 1; }
    FILE* out = std::fopen(argv[5], "wb");
    if (!out || std::fwrite(file.data(), 1, file.size(), out) != file.size()) { std::fprintf(stderr, "cannot write %s\n", argv[5]); return 1; }
    std::fclose(out);
    DestroyThreads();
    std::fprintf(stderr, "%s%s: %dx%d -> %zu bytes (%s)\n", slice_pixels ? "\n" : "", argv[5], desc.width, desc.height, file.size(), f->name);
    return 0;
}
