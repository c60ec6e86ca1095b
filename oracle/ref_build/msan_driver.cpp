// TEST INFRASTRUCTURE (study tool, tools/uninit_read_study.py): runs the reference's kernel.ispc (scalar build, see
// ispc_as_cpp/) under MemorySanitizer over every preset, so that EVERY read of storage the source leaves uninitialised
// that reaches a decision is reported with its kernel.ispc line and the line of the variable it came from
// (-fsanitize-memory-track-origins).  Built by `make -C oracle/ref_build msan` into oracle/_ref/ref_msan_uninit.
// Plain C I/O only: libstdc++ is not instrumented.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sanitizer/msan_interface.h>
#include "ispc_texcomp.h"

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// classes of 4x4 blocks: smooth, two-colour, noise, flat, translucent / opaque / mixed alpha
static void fill_ldr(uint8_t* img, int w, int h, uint32_t seed)
{
    for (int by = 0; by < h / 4; by++) for (int bx = 0; bx < w / 4; bx++) {
        const uint32_t cls = lcg(seed) % 6u;
        uint8_t a[4], b[4];
        for (int c = 0; c < 4; c++) { a[c] = (uint8_t)lcg(seed); b[c] = (uint8_t)lcg(seed); }
        const uint32_t amode = lcg(seed) % 3u;
        for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
            uint8_t* p = img + ((by * 4 + y) * w + bx * 4 + x) * 4;
            for (int c = 0; c < 4; c++) {
                int v;
                if (cls == 0) v = a[c] + (b[c] - a[c]) * (x + y) / 6;
                else if (cls == 1) v = ((lcg(seed) & 1u) ? a[c] : b[c]);
                else if (cls == 2) v = (int)(lcg(seed) & 255u);
                else if (cls == 3) v = a[c];
                else if (cls == 4) v = a[c] + (int)(lcg(seed) % 9u) - 4;
                else v = (x < 2 ? a[c] : b[c]) + (int)(lcg(seed) % 5u) - 2;
                p[c] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v);
            }
            if (amode == 0) p[3] = 255; else if (amode == 1 && (lcg(seed) & 3u)) p[3] = 255;
        }
    }
}
static void fill_hdr(uint16_t* img, int w, int h, uint32_t seed, int random_bits)
{
    for (int i = 0; i < w * h; i++) {
        for (int c = 0; c < 3; c++) img[i * 4 + c] = random_bits ? (uint16_t)lcg(seed) : (uint16_t)(0x3000u + (lcg(seed) % 0x1800u));
        img[i * 4 + 3] = 0x3C00;
    }
}

// Output blocks whose BYTES depend on uninitialised storage (their shadow is poisoned): count them, and let MSan print the
// origin chain of the first one (which variable, declared where, stored where).
static void audit(const char* what, const uint8_t* out, size_t nblocks, size_t bpb)
{
    size_t bad = 0, first = (size_t)-1;
    for (size_t b = 0; b < nblocks; b++)
        if (__msan_test_shadow(out + b * bpb, bpb) >= 0) { if (first == (size_t)-1) first = b; bad++; }
    printf("%-22s %6zu of %6zu output blocks depend on uninitialised storage\n", what, bad, nblocks);
    if (const char* dump = getenv("ITW_MSAN_DUMP")) {                   // the stream itself, to compare with the other builds
        FILE* f = fopen(dump, "wb");
        if (f) { __msan_unpoison(out, nblocks * bpb); fwrite(out, 1, nblocks * bpb, f); fclose(f); }
    }
    fflush(stdout);
    if (bad) {
        fprintf(stderr, "@@ output of %s, block %zu: origin of its uninitialised bytes\n", what, first);
        __msan_check_mem_is_initialized(out + first * bpb, bpb);
    }
}

int main(int argc, char** argv)
{
    const int w = 256, h = 128;
    uint8_t* ldr = (uint8_t*)malloc((size_t)w * h * 4);
    uint16_t* hdr = (uint16_t*)malloc((size_t)w * h * 8);
    uint8_t* out = (uint8_t*)malloc((size_t)(w / 4) * (h / 4) * 16);
    typedef void (*P7)(bc7_enc_settings*);
    typedef void (*P6)(bc6h_enc_settings*);
    const struct { const char* n; P7 f; } p7[] = {
        {"ultrafast", GetProfile_ultrafast}, {"veryfast", GetProfile_veryfast}, {"fast", GetProfile_fast}, {"basic", GetProfile_basic}, {"slow", GetProfile_slow},
        {"alpha_ultrafast", GetProfile_alpha_ultrafast}, {"alpha_veryfast", GetProfile_alpha_veryfast}, {"alpha_fast", GetProfile_alpha_fast},
        {"alpha_basic", GetProfile_alpha_basic}, {"alpha_slow", GetProfile_alpha_slow}};
    const struct { const char* n; P6 f; } p6[] = {
        {"bc6h_veryfast", GetProfile_bc6h_veryfast}, {"bc6h_fast", GetProfile_bc6h_fast}, {"bc6h_basic", GetProfile_bc6h_basic},
        {"bc6h_slow", GetProfile_bc6h_slow}, {"bc6h_veryslow", GetProfile_bc6h_veryslow}};
    const int only = argc > 1 ? atoi(argv[1]) : -1;                    // preset index 0..16, or all
    // optional: a raw RGBA8 file (tools/uninit_read_study.py dumps the golden photo) instead of the synthetic LDR surface
    uint8_t* file_img = nullptr; int fw = 0, fh = 0;
    if (argc > 4) {
        fw = atoi(argv[3]); fh = atoi(argv[4]);
        file_img = (uint8_t*)malloc((size_t)fw * fh * 4);
        FILE* f = fopen(argv[2], "rb");
        if (!f || fread(file_img, 1, (size_t)fw * fh * 4, f) != (size_t)fw * fh * 4) { fprintf(stderr, "cannot read %s\n", argv[2]); return 2; }
        fclose(f);
        free(out); out = (uint8_t*)malloc((size_t)(fw / 4) * (fh / 4) * 16);
    }
    for (uint32_t round = 0; round < (file_img ? 1u : 2u); round++) {
        fill_ldr(ldr, w, h, 12345u + 977u * round);
        rgba_surface s; s.ptr = ldr; s.width = w; s.height = h; s.stride = w * 4;
        if (file_img) { s.ptr = file_img; s.width = fw; s.height = fh; s.stride = fw * 4; }
        if (only < 0 || only == 15) { fprintf(stderr, "@@ preset bc1\n"); CompressBlocksBC1(&s, out); audit("bc1", out, (size_t)(s.width / 4) * (s.height / 4), 8); }
        if (only < 0 || only == 16) { fprintf(stderr, "@@ preset bc3\n"); CompressBlocksBC3(&s, out); audit("bc3", out, (size_t)(s.width / 4) * (s.height / 4), 16); }
        for (int i = 0; i < 10; i++) {
            if (only >= 0 && only != i) continue;
            bc7_enc_settings st;
            memset(&st, 0, sizeof st);                                   // the struct's own uninitialised slot (refineIterations[7] in RGB presets) is pinned to 0: SURVEY 8c S10
            p7[i].f(&st);
            fprintf(stderr, "@@ preset bc7_%s\n", p7[i].n);
            CompressBlocksBC7(&s, out, &st);
            char nm[64]; snprintf(nm, sizeof nm, "bc7_%s", p7[i].n);
            audit(nm, out, (size_t)(s.width / 4) * (s.height / 4), 16);
        }
        // BC6H: a small surface -- its (harmless, see tools/uninit_read_study.py) reads are reported per block and decision
        const int wh = 32, hh = 16;
        fill_hdr(hdr, wh, hh, 777u + round, (int)round);
        rgba_surface sh; sh.ptr = (uint8_t*)hdr; sh.width = wh; sh.height = hh; sh.stride = wh * 8;
        for (int i = 0; i < 5 && !file_img; i++) {
            if (only >= 0 && only != 10 + i) continue;
            bc6h_enc_settings st;
            memset(&st, 0, sizeof st);
            p6[i].f(&st);
            fprintf(stderr, "@@ preset %s\n", p6[i].n);
            CompressBlocksBC6H(&sh, out, &st);
            audit(p6[i].n, out, (size_t)(wh / 4) * (hh / 4), 16);
        }
    }
    return 0;
}
