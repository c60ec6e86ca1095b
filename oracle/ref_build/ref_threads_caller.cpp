/*
 * oracle/ref_build/ref_threads_caller.cpp -- TEST INFRASTRUCTURE.
 *
 * Drives the REFERENCE's own dispatch layer -- win32Threads.cpp compiled unmodified from /root/reference (through the
 * pthread Win32 shim in this directory) -- exactly as IntelPlugin::ISPC_compression does
 * (/root/reference/IntelCompressionPlugin/IntelPlugin.cpp:816-884): pick a CompressImage* trampoline, cut the surface
 * into 0x40000-pixel slices (restated below from IntelPlugin.cpp:851-879; that file needs the Photoshop SDK and
 * cannot be compiled), hand each slice to CompressImageMT (one band per worker thread, win32Threads.cpp:211-249) or
 * CompressImageST.  Linked twice: against the product library (…_gpu: the reference's caller on top of the gfx950
 * encoder, host pointers, up to 64 concurrent tiny calls) and against oracle/_ref/libispc_texcomp_ref.so (…_cpu).
 *
 *   ref_threads_caller <mt|st> <trampoline> <w> <h> <in.bin> <out.bin> [whole]
 *        trampoline = BC1 | BC3 | BC7_<profile> | BC6H_<profile>;  "whole" = one call instead of the slice loop
 *        worker count = host cores, or ITW_REF_THREADS;  ITW_REF_REPS=n repeats the whole image n times and reports
 *        the fastest pass (the first pass of a process pays HIP start-up: runtime init, code-object load, first hipMalloc)
 */
#include "win32Threads.h"      /* the reference's header */
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define T(n) {#n, CompressImage##n}
static const struct { const char* name; CompressionFunc* fn; } kTramp[] = {
    T(BC1), T(BC3),
    T(BC7_ultrafast), T(BC7_veryfast), T(BC7_fast), T(BC7_basic), T(BC7_slow),
    T(BC7_alpha_ultrafast), T(BC7_alpha_veryfast), T(BC7_alpha_fast), T(BC7_alpha_basic), T(BC7_alpha_slow),
    T(BC6H_veryfast), T(BC6H_fast), T(BC6H_basic), T(BC6H_slow), T(BC6H_veryslow)};
#undef T

int main(int argc, char** argv)
{
    if (argc < 7) { fprintf(stderr, "usage: %s <mt|st> <trampoline> <w> <h> <in> <out> [whole]\n", argv[0]); return 2; }
    bool mt = !strcmp(argv[1], "mt"), whole = argc > 7 && !strcmp(argv[7], "whole");
    std::string name = argv[2];
    CompressionFunc* fn = nullptr;
    for (auto& t : kTramp) if (name == t.name) fn = t.fn;
    if (!fn) { fprintf(stderr, "unknown trampoline %s\n", argv[2]); return 2; }
    DXGI_FORMAT fmt = name == "BC1" ? DXGI_FORMAT_BC1_UNORM : name == "BC3" ? DXGI_FORMAT_BC3_UNORM
                    : name.rfind("BC7", 0) == 0 ? DXGI_FORMAT_BC7_UNORM : DXGI_FORMAT_BC6H_UF16;
    int w = atoi(argv[3]), h = atoi(argv[4]);
    int bpp = fmt == DXGI_FORMAT_BC6H_UF16 ? 8 : 4, bpb = fmt == DXGI_FORMAT_BC1_UNORM ? 8 : 16;

    FILE* f = fopen(argv[5], "rb");
    if (!f) { perror(argv[5]); return 2; }
    std::vector<uint8_t> in((size_t)w * h * bpp);
    if (fread(in.data(), 1, in.size(), f) != in.size()) { fprintf(stderr, "short input\n"); return 2; }
    fclose(f);
    std::vector<uint8_t> out((size_t)(w / 4) * (h / 4) * bpb, 0xEE);
    size_t row_pitch = (size_t)(w / 4) * bpb;           /* DirectXTex rowPitch of a BC image = block row pitch */

    rgba_surface source;
    source.ptr = in.data(); source.width = w; source.height = h; source.stride = w * bpp;

    if (mt) InitWin32Threads();
    int reps = getenv("ITW_REF_REPS") ? atoi(getenv("ITW_REF_REPS")) : 1;
    if (reps < 1) reps = 1;
    double ms = 1e30, first_ms = 0;
    int slices = whole ? 1 : (source.width * source.height) / 0x40000;         /* IntelPlugin.cpp:851 */
    if (slices < 1) slices = 1;
    for (int rep = 0; rep < reps; rep++) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < slices; i++) {
        int ylo = (int)((long long)i * source.height / slices) & ~3;          /* IntelPlugin.cpp:860-861 */
        int yhi = (int)((long long)(i + 1) * source.height / slices) & ~3;
        if (yhi > source.height) yhi = source.height;
        if (yhi <= ylo) continue;
        rgba_surface input = source;
        input.ptr += (size_t)input.stride * ylo;
        input.height = yhi - ylo;
        uint8_t* dst = out.data() + row_pitch * (size_t)(ylo >> 2);
        if (mt) CompressImageMT(&input, dst, fn, fmt);
        else    CompressImageST(&input, dst, fn, fmt);
    }
    double pass = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    if (rep == 0) first_ms = pass;
    if (pass < ms) ms = pass;
    }
    int workers = mt ? GetProcessorCount() : 1;
    if (mt) DestroyThreads();
    printf("{\"trampoline\": \"%s\", \"mode\": \"%s\", \"slices\": %d, \"workers\": %d, \"reps\": %d, \"first_pass_ms\": %.3f, "
           "\"ms\": %.3f, \"mpix_s\": %.1f}\n", argv[2], argv[1], slices, workers, reps, first_ms, ms, (double)w * h / ms / 1e3);

    f = fopen(argv[6], "wb");
    if (!f || fwrite(out.data(), 1, out.size(), f) != out.size()) { perror(argv[6]); return 2; }
    fclose(f);
    return 0;
}
