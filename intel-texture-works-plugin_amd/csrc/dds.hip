// dds.hip -- DDS container for BCn block streams (include/itw_dds.h); host code only.
// Header contents follow DirectXTex's _EncodeDDSHeader for the formats the plugin saves
// (DirectXTexDDS.cpp:441-675; constants DDS.h:38-236); data order as SaveToDDSMemory (:1611-1700):
// for each array item / cube face, its mip chain from the top level down, tightly packed.
#include <cstring>
#include "../../include/itw_dds.h"

namespace {

constexpr uint32_t DDS_MAGIC = 0x20534444;                 // "DDS "
constexpr uint32_t DDS_FOURCC = 0x4;
constexpr uint32_t HEADER_FLAGS_TEXTURE = 0x00001007, HEADER_FLAGS_MIPMAP = 0x00020000, HEADER_FLAGS_LINEARSIZE = 0x00080000;
constexpr uint32_t SURFACE_FLAGS_TEXTURE = 0x00001000, SURFACE_FLAGS_MIPMAP = 0x00400008, SURFACE_FLAGS_CUBEMAP = 0x00000008;
constexpr uint32_t CUBEMAP_ALLFACES = 0x0000fe00;
constexpr uint32_t DIMENSION_TEXTURE2D = 3, MISC_TEXTURECUBE = 0x4;

constexpr uint32_t fourcc(char a, char b, char c, char d)
{
    return (uint32_t)(uint8_t)a | ((uint32_t)(uint8_t)b << 8) | ((uint32_t)(uint8_t)c << 16) | ((uint32_t)(uint8_t)d << 24);
}

int bytes_per_block(uint32_t f)
{
    switch (f) {
    case 71: case 72: case 80: return 8;                    // BC1, BC4_UNORM
    case 77: case 78: case 83: case 95: case 96: case 98: case 99: return 16;   // BC3, BC5_UNORM, BC6H, BC7
    default: return 0;
    }
}

// legacy FourCC for the formats DirectXTex maps without the DX10 extension
// (DirectXTexDDS.cpp:474-483; DDS.h:71-90: BC4_UNORM -> "BC4U", BC5_UNORM -> "BC5U")
uint32_t legacy_fourcc(uint32_t f)
{
    switch (f) {
    case 71: return fourcc('D', 'X', 'T', '1');
    case 77: return fourcc('D', 'X', 'T', '5');
    case 80: return fourcc('B', 'C', '4', 'U');
    case 83: return fourcc('B', 'C', '5', 'U');
    default: return 0u;
    }
}

bool needs_dx10(const ItwDdsDesc& d)
{
    // arrays other than a single cube need the extension (DirectXTexDDS.cpp:450-457)
    const uint32_t items = d.array_size ? d.array_size : 1;
    return legacy_fourcc(d.dxgi_format) == 0 || items > 1;
}

void put32(uint8_t* p, size_t word, uint32_t v) { std::memcpy(p + 4 * word, &v, 4); }
uint32_t get32(const uint8_t* p, size_t word) { uint32_t v; std::memcpy(&v, p + 4 * word, 4); return v; }

} // namespace

extern "C" {

size_t itwDdsLevelBytes(uint32_t f, uint32_t w, uint32_t h)
{
    const size_t nbw = (w + 3) / 4 ? (w + 3) / 4 : 1, nbh = (h + 3) / 4 ? (h + 3) / 4 : 1;
    return nbw * nbh * (size_t)bytes_per_block(f);
}

size_t itwDdsHeaderBytes(const ItwDdsDesc* d)
{
    if (!d || !bytes_per_block(d->dxgi_format) || !d->width || !d->height || !d->mip_levels) return 0;
    return 4 + 124 + (needs_dx10(*d) ? 20 : 0);
}

size_t itwDdsFileBytes(const ItwDdsDesc* d)
{
    const size_t hdr = itwDdsHeaderBytes(d);
    if (!hdr) return 0;
    size_t chain = 0;
    uint32_t w = d->width, h = d->height;
    for (uint32_t m = 0; m < d->mip_levels; m++) {
        chain += itwDdsLevelBytes(d->dxgi_format, w, h);
        w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1;
    }
    const size_t items = (size_t)(d->array_size ? d->array_size : 1) * (d->is_cubemap ? 6 : 1);
    return hdr + chain * items;
}

size_t itwDdsWriteHeader(const ItwDdsDesc* d, uint8_t* dst, size_t capacity)
{
    const size_t need = itwDdsHeaderBytes(d);
    if (!need || !dst || capacity < need) return 0;
    std::memset(dst, 0, need);
    put32(dst, 0, DDS_MAGIC);
    uint8_t* h = dst + 4;                                   // DDS_HEADER, 31 dwords
    put32(h, 0, 124);                                       // dwSize
    put32(h, 1, HEADER_FLAGS_TEXTURE | HEADER_FLAGS_MIPMAP | HEADER_FLAGS_LINEARSIZE);   // mipLevels > 0 always sets MIPMAP (:553)
    put32(h, 2, d->height);
    put32(h, 3, d->width);
    put32(h, 4, (uint32_t)itwDdsLevelBytes(d->dxgi_format, d->width, d->height));         // slice pitch of the top level
    put32(h, 5, 1);                                         // dwDepth = 1 for 2D (:586)
    put32(h, 6, d->mip_levels);
    uint32_t caps = SURFACE_FLAGS_TEXTURE, caps2 = 0;
    if (d->mip_levels > 1) caps |= SURFACE_FLAGS_MIPMAP;
    if (d->is_cubemap) { caps |= SURFACE_FLAGS_CUBEMAP; caps2 |= CUBEMAP_ALLFACES; }
    // ddspf at dwords 18..25
    put32(h, 18, 32);
    put32(h, 19, DDS_FOURCC);
    put32(h, 20, needs_dx10(*d) ? fourcc('D', 'X', '1', '0') : legacy_fourcc(d->dxgi_format));
    put32(h, 26, caps);
    put32(h, 27, caps2);
    if (needs_dx10(*d)) {
        uint8_t* e = h + 124;                               // DDS_HEADER_DXT10
        put32(e, 0, d->dxgi_format);
        put32(e, 1, DIMENSION_TEXTURE2D);
        put32(e, 2, d->is_cubemap ? MISC_TEXTURECUBE : 0);
        put32(e, 3, d->array_size ? d->array_size : 1);     // cubes: number of cubes (:646-650)
        put32(e, 4, 0);                                     // miscFlags2 stays 0 unless forced (:664-668)
    }
    return need;
}

size_t itwDdsReadHeader(const uint8_t* src, size_t size, ItwDdsDesc* out)
{
    if (!src || !out || size < 128 || get32(src, 0) != DDS_MAGIC) return 0;
    const uint8_t* h = src + 4;
    if (get32(h, 0) != 124 || get32(h, 18) != 32 || !(get32(h, 19) & DDS_FOURCC)) return 0;
    ItwDdsDesc d;
    d.height = get32(h, 2); d.width = get32(h, 3);
    d.mip_levels = get32(h, 6) ? get32(h, 6) : 1;
    d.is_cubemap = (get32(h, 27) & 0x200) ? 1 : 0;
    d.array_size = 1;
    size_t off = 128;
    const uint32_t cc = get32(h, 20);
    if (cc == fourcc('D', 'X', 'T', '1')) d.dxgi_format = 71;
    else if (cc == fourcc('D', 'X', 'T', '5')) d.dxgi_format = 77;
    else if (cc == fourcc('B', 'C', '4', 'U') || cc == fourcc('A', 'T', 'I', '1')) d.dxgi_format = 80;   // DirectXTexDDS.cpp:64-70
    else if (cc == fourcc('B', 'C', '5', 'U') || cc == fourcc('A', 'T', 'I', '2')) d.dxgi_format = 83;
    else if (cc == fourcc('D', 'X', '1', '0')) {
        if (size < 148) return 0;
        const uint8_t* e = h + 124;
        d.dxgi_format = get32(e, 0);
        if (get32(e, 1) != DIMENSION_TEXTURE2D) return 0;
        d.is_cubemap = (get32(e, 2) & MISC_TEXTURECUBE) ? 1 : 0;
        d.array_size = get32(e, 3) ? get32(e, 3) : 1;
        off = 148;
    } else return 0;
    if (!bytes_per_block(d.dxgi_format) || !d.width || !d.height) return 0;
    *out = d;
    return off;
}

size_t itwDdsWriteFile(const ItwDdsDesc* d, const uint8_t* const* levels, size_t nlevels, uint8_t* dst, size_t capacity)
{
    const size_t total = itwDdsFileBytes(d);
    if (!total || !levels || !dst || capacity < total) return 0;
    const size_t items = (size_t)(d->array_size ? d->array_size : 1) * (d->is_cubemap ? 6 : 1);
    if (nlevels != items * d->mip_levels) return 0;
    size_t pos = itwDdsWriteHeader(d, dst, capacity);
    if (!pos) return 0;
    size_t idx = 0;
    for (size_t it = 0; it < items; it++) {
        uint32_t w = d->width, h = d->height;
        for (uint32_t m = 0; m < d->mip_levels; m++) {
            const size_t n = itwDdsLevelBytes(d->dxgi_format, w, h);
            if (!levels[idx]) return 0;
            std::memcpy(dst + pos, levels[idx], n);
            pos += n; idx++;
            w = w > 1 ? w / 2 : 1; h = h > 1 ? h / 2 : 1;
        }
    }
    return pos;
}

} // extern "C"
