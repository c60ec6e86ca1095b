/*
 * oracle/ref_build/ref_header_caller.cpp -- TEST INFRASTRUCTURE.
 *
 * A caller compiled against the REFERENCE's own public header
 * (/root/reference/3rdParty/Intel/Source/ispc_texcomp.h:19-107, found through -I, never copied) and linked against
 * whichever library provides that ABI: the product (intel-texture-works-plugin_amd/lib/libispc_texcomp.so -> binary
 * ref_header_caller_gpu) or oracle/_ref/libispc_texcomp_ref.so (-> ref_header_caller_cpu).  If the product's
 * structs, symbol names or calling convention drifted from the reference header, this TU would fail to link or the
 * bytes it writes would differ from the oracle's.
 *
 *   ref_header_caller profiles <out.bin>                      15 settings structs over 0xA5-filled storage
 *   ref_header_caller encode <fmt> <profile|-> <w> <h> <in.bin> <out.bin>
 *        fmt = bc1 | bc3 | bc7 | bc6h ; in.bin = tightly packed RGBA8 (RGBA16F bit patterns for bc6h)
 */
#include "ispc_texcomp.h"      /* the reference's header */
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef void (*bc7_profile_fn)(bc7_enc_settings*);
typedef void (*bc6h_profile_fn)(bc6h_enc_settings*);
static const struct { const char* name; bc7_profile_fn fn; } kBc7[] = {
    {"ultrafast", GetProfile_ultrafast}, {"veryfast", GetProfile_veryfast}, {"fast", GetProfile_fast},
    {"basic", GetProfile_basic}, {"slow", GetProfile_slow},
    {"alpha_ultrafast", GetProfile_alpha_ultrafast}, {"alpha_veryfast", GetProfile_alpha_veryfast},
    {"alpha_fast", GetProfile_alpha_fast}, {"alpha_basic", GetProfile_alpha_basic}, {"alpha_slow", GetProfile_alpha_slow}};
static const struct { const char* name; bc6h_profile_fn fn; } kBc6h[] = {
    {"veryfast", GetProfile_bc6h_veryfast}, {"fast", GetProfile_bc6h_fast}, {"basic", GetProfile_bc6h_basic},
    {"slow", GetProfile_bc6h_slow}, {"veryslow", GetProfile_bc6h_veryslow}};

static std::vector<uint8_t> slurp(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> v((size_t)n);
    if (fread(v.data(), 1, v.size(), f) != v.size()) { perror("fread"); exit(2); }
    fclose(f);
    return v;
}

static void spill(const char* path, const void* p, size_t n)
{
    FILE* f = fopen(path, "wb");
    if (!f || fwrite(p, 1, n, f) != n) { perror(path); exit(2); }
    fclose(f);
}

int main(int argc, char** argv)
{
    if (argc == 3 && !strcmp(argv[1], "profiles")) {
        std::vector<uint8_t> out;
        for (auto& p : kBc7) {
            alignas(8) uint8_t raw[sizeof(bc7_enc_settings)];
            memset(raw, 0xA5, sizeof raw);
            p.fn(reinterpret_cast<bc7_enc_settings*>(raw));
            out.insert(out.end(), raw, raw + sizeof raw);
        }
        for (auto& p : kBc6h) {
            alignas(8) uint8_t raw[sizeof(bc6h_enc_settings)];
            memset(raw, 0xA5, sizeof raw);
            p.fn(reinterpret_cast<bc6h_enc_settings*>(raw));
            out.insert(out.end(), raw, raw + sizeof raw);
        }
        spill(argv[2], out.data(), out.size());
        return 0;
    }
    if (argc == 8 && !strcmp(argv[1], "encode")) {
        std::string fmt = argv[2], prof = argv[3];
        int w = atoi(argv[4]), h = atoi(argv[5]);
        std::vector<uint8_t> in = slurp(argv[6]);
        int bpp = fmt == "bc6h" ? 8 : 4;
        if ((size_t)w * h * bpp != in.size()) { fprintf(stderr, "input size mismatch\n"); return 2; }
        rgba_surface s;
        s.ptr = in.data(); s.width = w; s.height = h; s.stride = w * bpp;
        size_t bpb = fmt == "bc1" ? 8 : 16;
        std::vector<uint8_t> out((size_t)(w / 4) * (h / 4) * bpb, 0xEE);
        if (fmt == "bc1") CompressBlocksBC1(&s, out.data());
        else if (fmt == "bc3") CompressBlocksBC3(&s, out.data());
        else if (fmt == "bc7") {
            bc7_enc_settings st;
            memset(&st, 0, sizeof st);
            bool ok = false;
            for (auto& p : kBc7) if (prof == p.name) { p.fn(&st); ok = true; }
            if (!ok) { fprintf(stderr, "unknown bc7 profile\n"); return 2; }
            CompressBlocksBC7(&s, out.data(), &st);
        } else if (fmt == "bc6h") {
            bc6h_enc_settings st;
            memset(&st, 0, sizeof st);
            bool ok = false;
            for (auto& p : kBc6h) if (prof == p.name) { p.fn(&st); ok = true; }
            if (!ok) { fprintf(stderr, "unknown bc6h profile\n"); return 2; }
            CompressBlocksBC6H(&s, out.data(), &st);
        } else { fprintf(stderr, "unknown format\n"); return 2; }
        spill(argv[7], out.data(), out.size());
        return 0;
    }
    fprintf(stderr, "usage: %s profiles <out> | encode <fmt> <profile|-> <w> <h> <in> <out>\n", argv[0]);
    return 2;
}
