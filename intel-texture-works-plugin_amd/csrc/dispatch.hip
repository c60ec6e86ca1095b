// dispatch.hip -- portable, GPU-aware restatement of the reference's dispatch layer (include/itw_dispatch.h):
// CompressImageMT/ST and the profile trampolines (win32Threads.cpp:192-329), the plugin's slice loop with progress /
// early out (IntelPlugin.cpp:851-879) and the pad-to-multiple-of-4 pre-pass (IntelPlugin.cpp:893-928), the latter
// also as a device kernel so a GPU-resident pipeline never leaves HBM.
//
// Workers are GPUs, not CPU cores: one persistent host thread per visible device, each bound to its device once.
// A band is handed to a worker exactly like win32Threads.cpp:217-231 cuts them, so with host pointers every GPU
// stages and encodes its own band concurrently (H2D/D2H of different devices overlap on their own PCIe links);
// a device-resident surface is encoded by the device that owns it, in one call.
#include <hip/hip_runtime.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/itw_dispatch.h"
#include "../../include/itw_amd.h"
#include "../../include/itw_bc45.h"
#include "host_rt.hpp"

namespace {
using itw::is_device_pointer;       // one predicate library-wide: device AND managed memory count as resident

struct Job {
    rgba_surface input;
    uint8_t* output = nullptr;
    CompressionFunc* fn = nullptr;
    int min_height = 4;         // bands without a whole block row are not handed out (the reference calls them anyway,
                                // win32Threads.cpp:264, and its kernel loops over height/4 = 0 rows); 1 for BC4/BC5
    std::function<void(int)> task;   // instead of a band: something the worker runs on its device (its share of a pipelined slice loop)
    bool pending = false;
};

struct Pool {
    std::mutex m;
    std::condition_variable work, done;
    std::vector<std::thread> threads;
    std::vector<Job> jobs;
    int outstanding = 0;
    bool quit = false;
    bool failed = false;        // a worker's call failed (error mode "return"): message in fail_msg
    char fail_msg[384] = {0};
    std::mutex submit;          // one CompressImageMT at a time, like the reference's single global pool

    int devices = 1;

    void run(int idx)
    {
        (void)hipSetDevice(idx % devices);
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            work.wait(lk, [&] { return quit || jobs[idx].pending; });
            if (quit) return;
            Job j = jobs[idx];
            lk.unlock();
            const char* err = nullptr;
            if (j.task) {
                itwClearError();
                j.task(idx);
                err = itwLastError();
            } else if (j.input.height >= j.min_height) {
                itwClearError();
                j.fn(&j.input, j.output);
                // The resident path of CompressBlocks* is asynchronous on this worker's stream; a band handed to a
                // worker is only done when its kernels are (managed surfaces can take this route through callers
                // that bypass CompressImageMT's own resident shortcut).
                if (is_device_pointer(j.input.ptr) && is_device_pointer(j.output)) (void)hipStreamSynchronize((hipStream_t)itwGetStream());
                err = itwLastError();
            }
            lk.lock();
            if (err && !failed) { failed = true; std::snprintf(fail_msg, sizeof fail_msg, "%s", err); }
            jobs[idx].pending = false;
            if (--outstanding == 0) done.notify_all();
        }
    }
};

Pool* g_pool = nullptr;
std::mutex g_pool_mutex;

int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) n = 1;
    return n > 64 ? 64 : n;     // kMaxWinThreads
}

// Workers = devices, unless ITW_WORKERS=<n> asks for more: the extra workers share the devices round-robin.  That is how
// the 8-band path is exercised on a single-GPU box (tests), and it lets a host overlap staging of one band with the
// encode of another on one device.
int worker_count()
{
    const int dev = device_count();
    const char* e = std::getenv("ITW_WORKERS");
    int n = e ? std::atoi(e) : 0;
    if (n < 1) n = dev;
    return n > 64 ? 64 : n;
}

Pool* pool()
{
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (!g_pool) {
        Pool* p = new Pool;
        const int n = worker_count();
        p->devices = device_count();
        p->jobs.resize(n);
        for (int i = 0; i < n; i++) p->threads.emplace_back([p, i] { p->run(i); });
        g_pool = p;
    }
    return g_pool;
}

template <int PIXEL_WORDS>
__global__ void pad_kernel(const uint32_t* __restrict__ src, int64_t src_stride_words, int w, int h,
                           uint32_t* __restrict__ dst, int ow, int oh)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= ow || y >= oh) return;
    const int sx = min(x, w - 1), sy = min(y, h - 1);          // edge replication
    const uint32_t* s = src + (int64_t)sy * src_stride_words + (int64_t)sx * PIXEL_WORDS;
    uint32_t* d = dst + ((int64_t)y * ow + x) * PIXEL_WORDS;
#pragma unroll
    for (int i = 0; i < PIXEL_WORDS; i++) d[i] = s[i];
}

} // namespace

extern "C" {

int GetProcessorCount(void) { return worker_count(); }

void InitWin32Threads(void) { (void)pool(); }

void DestroyThreads(void)
{
    std::lock_guard<std::mutex> lk(g_pool_mutex);
    if (!g_pool) return;
    {
        std::lock_guard<std::mutex> l2(g_pool->m);
        g_pool->quit = true;
    }
    g_pool->work.notify_all();
    for (auto& t : g_pool->threads) t.join();
    delete g_pool;
    g_pool = nullptr;
}

// The DirectXTex formats keep partial blocks (itw_bc45.h); the ISPC formats drop them (kernel.ispc:600-601).
static bool keeps_partial_blocks(int f) { return f == ITW_DXGI_FORMAT_BC4_UNORM || f == ITW_DXGI_FORMAT_BC5_UNORM; }
static int blocks_across(int width, int f) { return keeps_partial_blocks(f) ? (width + 3) / 4 : width / 4; }

int GetBytesPerBlock(int f)
{
    switch (f) {
    case ITW_DXGI_FORMAT_BC5_UNORM:                              // not in the reference's switch (BC5 never reaches it there)
    case ITW_DXGI_FORMAT_BC3_UNORM_SRGB: case ITW_DXGI_FORMAT_BC3_UNORM:
    case ITW_DXGI_FORMAT_BC7_UNORM_SRGB: case ITW_DXGI_FORMAT_BC7_UNORM:
    case ITW_DXGI_FORMAT_BC6H_UF16: case ITW_DXGI_FORMAT_BC6H_SF16:
        return 16;
    default:
        return 8;
    }
}

bool CompressImageST(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int)
{
    itwClearError();
    (*cmpFunc)(input, output);
    return itwLastError() == nullptr;         // always true in the default (abort) error mode, like the reference
}

bool CompressImageMT(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format)
{
    Pool* p = pool();
    const int n = (int)p->threads.size();
    // a surface that already lives on one GPU is encoded there, in one call
    if (n == 1 || is_device_pointer(input->ptr) || is_device_pointer(output)) return CompressImageST(input, output, cmpFunc, dxgi_format);

    const int bpb = GetBytesPerBlock(dxgi_format);
    const int lines = (input->height + n - 1) / n;               // win32Threads.cpp:217
    std::lock_guard<std::mutex> one(p->submit);
    {
        std::lock_guard<std::mutex> lk(p->m);
        for (int i = 0; i < n; i++) {
            int y0 = (lines * i) / 4 * 4, y1 = (lines * (i + 1)) / 4 * 4;
            if (y1 > input->height) y1 = input->height;
            if (i == n - 1 && keeps_partial_blocks(dxgi_format)) y1 = input->height;   // the partial block row, if any
            Job& j = p->jobs[i];
            j.input = *input;
            j.input.ptr = input->ptr + (int64_t)y0 * input->stride;
            j.input.height = y1 - y0;
            j.output = output + (int64_t)(y0 / 4) * blocks_across(input->width, dxgi_format) * bpb;
            j.fn = cmpFunc;
            j.task = nullptr;
            j.min_height = keeps_partial_blocks(dxgi_format) ? 1 : 4;
            j.pending = true;
        }
        p->outstanding = n;
        p->failed = false;
    }
    p->work.notify_all();
    std::unique_lock<std::mutex> lk(p->m);
    p->done.wait(lk, [&] { return p->outstanding == 0; });
    if (p->failed) {
        itw::Failure f;
        std::snprintf(f.msg, sizeof f.msg, "%s", p->fail_msg);
        lk.unlock();
        itw::report_failure(f);               // surfaces the worker's message on the calling thread
        return false;
    }
    return true;
}

void CompressImageBC1(const rgba_surface* input, uint8_t* output) { CompressBlocksBC1(input, output); }
void CompressImageBC3(const rgba_surface* input, uint8_t* output) { CompressBlocksBC3(input, output); }
void CompressImageBC4(const rgba_surface* input, uint8_t* output) { CompressBlocksBC4(input, output); }
void CompressImageBC5(const rgba_surface* input, uint8_t* output) { CompressBlocksBC5(input, output); }

#define ITW_BC7_TRAMPOLINE(profile)                                                    \
    void CompressImageBC7_##profile(const rgba_surface* input, uint8_t* output)        \
    {                                                                                  \
        bc7_enc_settings settings;                                                     \
        std::memset(&settings, 0, sizeof settings);                                    \
        GetProfile_##profile(&settings);                                               \
        CompressBlocksBC7(input, output, &settings);                                   \
    }
#define ITW_BC6H_TRAMPOLINE(profile)                                                   \
    void CompressImageBC6H_##profile(const rgba_surface* input, uint8_t* output)       \
    {                                                                                  \
        bc6h_enc_settings settings;                                                    \
        std::memset(&settings, 0, sizeof settings);                                    \
        GetProfile_bc6h_##profile(&settings);                                          \
        CompressBlocksBC6H(input, output, &settings);                                  \
    }
ITW_BC7_TRAMPOLINE(ultrafast) ITW_BC7_TRAMPOLINE(veryfast) ITW_BC7_TRAMPOLINE(fast) ITW_BC7_TRAMPOLINE(basic) ITW_BC7_TRAMPOLINE(slow)
ITW_BC7_TRAMPOLINE(alpha_ultrafast) ITW_BC7_TRAMPOLINE(alpha_veryfast) ITW_BC7_TRAMPOLINE(alpha_fast)
ITW_BC7_TRAMPOLINE(alpha_basic) ITW_BC7_TRAMPOLINE(alpha_slow)
ITW_BC6H_TRAMPOLINE(veryfast) ITW_BC6H_TRAMPOLINE(fast) ITW_BC6H_TRAMPOLINE(basic) ITW_BC6H_TRAMPOLINE(slow) ITW_BC6H_TRAMPOLINE(veryslow)

// The library's own trampolines are recognised by address: for them the slice loop runs as a pipeline (abi.hip, compress_sliced) with the
// format and the preset's settings in hand.  Any other CompressionFunc is opaque -- it can only be called, synchronously, slice by slice.
static bool resolve_trampoline(CompressionFunc* fn, int dxgi_format, bc7_enc_settings* s7, bc6h_enc_settings* s6, const void** settings)
{
    *settings = nullptr;
    const bool bc7 = dxgi_format == ITW_DXGI_FORMAT_BC7_UNORM || dxgi_format == ITW_DXGI_FORMAT_BC7_UNORM_SRGB;
    const bool bc6 = dxgi_format == ITW_DXGI_FORMAT_BC6H_UF16 || dxgi_format == ITW_DXGI_FORMAT_BC6H_SF16;
    if (fn == &CompressImageBC1) return dxgi_format == ITW_DXGI_FORMAT_BC1_UNORM || dxgi_format == ITW_DXGI_FORMAT_BC1_UNORM_SRGB;
    if (fn == &CompressImageBC3) return dxgi_format == ITW_DXGI_FORMAT_BC3_UNORM || dxgi_format == ITW_DXGI_FORMAT_BC3_UNORM_SRGB;
    if (fn == &CompressImageBC4) return dxgi_format == ITW_DXGI_FORMAT_BC4_UNORM;
    if (fn == &CompressImageBC5) return dxgi_format == ITW_DXGI_FORMAT_BC5_UNORM;
    struct P7 { CompressionFunc* fn; void (*get)(bc7_enc_settings*); };
    static const P7 p7[] = {
        {&CompressImageBC7_ultrafast, &GetProfile_ultrafast}, {&CompressImageBC7_veryfast, &GetProfile_veryfast}, {&CompressImageBC7_fast, &GetProfile_fast},
        {&CompressImageBC7_basic, &GetProfile_basic}, {&CompressImageBC7_slow, &GetProfile_slow},
        {&CompressImageBC7_alpha_ultrafast, &GetProfile_alpha_ultrafast}, {&CompressImageBC7_alpha_veryfast, &GetProfile_alpha_veryfast},
        {&CompressImageBC7_alpha_fast, &GetProfile_alpha_fast}, {&CompressImageBC7_alpha_basic, &GetProfile_alpha_basic},
        {&CompressImageBC7_alpha_slow, &GetProfile_alpha_slow}};
    for (const P7& e : p7)
        if (fn == e.fn) { if (!bc7) return false; std::memset(s7, 0, sizeof *s7); e.get(s7); *settings = s7; return true; }
    struct P6 { CompressionFunc* fn; void (*get)(bc6h_enc_settings*); };
    static const P6 p6[] = {
        {&CompressImageBC6H_veryfast, &GetProfile_bc6h_veryfast}, {&CompressImageBC6H_fast, &GetProfile_bc6h_fast}, {&CompressImageBC6H_basic, &GetProfile_bc6h_basic},
        {&CompressImageBC6H_slow, &GetProfile_bc6h_slow}, {&CompressImageBC6H_veryslow, &GetProfile_bc6h_veryslow}};
    for (const P6& e : p6)
        if (fn == e.fn) { if (!bc6) return false; std::memset(s6, 0, sizeof *s6); e.get(s6); *settings = s6; return true; }
    return false;
}

// Several GPUs, host memory: one PIPELINE PER GPU.  The windows of the slice loop are dealt round-robin to the pool's workers (window k to
// worker k % n; every worker runs its share through its own device, streams and staging: itw::sliced_part), a worker reports a window once
// its bytes are in `target`, and the submitting thread -- this one -- calls `progress` for the window's slices as soon as every window
// before it has arrived, i.e. in the reference's order.  A false return raises the stop flag: no worker issues another window, what is in
// flight is drained (windows other GPUs had already finished further down the image stay written, like the rest of the aborting window).
static bool sliced_on_pool(const rgba_surface* source, uint8_t* target, int dxgi_format, const void* settings, int64_t slice_pixels, int slices,
                           ItwProgressFunc* progress, void* user)
{
    Pool* p = pool();
    const int n = (int)p->threads.size();
    int W = 1;
    const int nwin = itw::sliced_windows(dxgi_format, settings, source->width, source->height, slice_pixels, &W);
    struct Shared {
        std::mutex m; std::condition_variable cv;
        std::vector<char> arrived;
        std::atomic<bool> stop{false};
        int running = 0, W = 1;
    } sh;
    sh.arrived.assign((size_t)(nwin > 0 ? nwin : 1), 0);
    sh.running = n; sh.W = W;
    auto retired = [](int s0, int, void* ctx) {
        Shared* s = static_cast<Shared*>(ctx);
        { std::lock_guard<std::mutex> lk(s->m); s->arrived[(size_t)(s0 / s->W)] = 1; }
        s->cv.notify_all();
    };
    std::lock_guard<std::mutex> one(p->submit);
    {
        std::lock_guard<std::mutex> lk(p->m);
        for (int i = 0; i < n; i++) {
            Job& j = p->jobs[i];
            j.fn = nullptr;
            j.task = [&, n](int idx) {
                itw::SlicedPart part;
                part.part = idx; part.parts = n; part.retired = retired; part.stop = &sh.stop; part.ctx = &sh;
                const bool ok = itw::sliced_part(source, target, dxgi_format, settings, slice_pixels, part);
                if (!ok) sh.stop.store(true, std::memory_order_release);          // a failed worker ends the job for everybody
                { std::lock_guard<std::mutex> l2(sh.m); sh.running--; }
                sh.cv.notify_all();
            };
            j.pending = true;
        }
        p->outstanding = n;
        p->failed = false;
    }
    p->work.notify_all();
    bool aborted = false;
    {
        std::unique_lock<std::mutex> lk(sh.m);
        for (int k = 0; k < nwin && !aborted; k++) {
            sh.cv.wait(lk, [&] { return sh.arrived[(size_t)k] || sh.running == 0; });
            if (!sh.arrived[(size_t)k]) break;                                      // the workers are gone without it: stopped or failed
            lk.unlock();
            const int s1 = (k + 1) * W < slices ? (k + 1) * W : slices;
            for (int i = k * W + 1; i <= s1 && i < slices; i++)
                if (progress && !progress(i, slices, user)) { aborted = true; sh.stop.store(true, std::memory_order_release); break; }
            lk.lock();
        }
    }
    bool failed = false;
    {
        std::unique_lock<std::mutex> lk(p->m);
        p->done.wait(lk, [&] { return p->outstanding == 0; });
        failed = p->failed;
        if (failed) {
            itw::Failure f;
            std::snprintf(f.msg, sizeof f.msg, "%s", p->fail_msg);
            lk.unlock();
            itw::report_failure(f);
        }
        for (int i = 0; i < n; i++) p->jobs[i].task = nullptr;                      // (the closures point into this frame)
    }
    return !aborted && !failed && !sh.stop.load();
}

bool itwCompressImageSliced(const rgba_surface* source, uint8_t* target, int64_t block_row_pitch, CompressionFunc* cmpFunc,
                            int dxgi_format, bool multithreaded, int64_t slice_pixels, ItwProgressFunc* progress, void* user)
{
    if (!cmpFunc) return false;
    if (slice_pixels <= 0) slice_pixels = 0x40000;                                   // IntelPlugin.cpp:851
    int slices = (int)(((int64_t)source->width * source->height) / slice_pixels);
    if (slices < 1) slices = 1;
    // the ABI packs block rows tightly; a wider pitch would need one call per block row
    const int64_t tight = (int64_t)blocks_across(source->width, dxgi_format) * GetBytesPerBlock(dxgi_format);
    if (block_row_pitch != tight) {
        std::fprintf(stderr, "itwCompressImageSliced: block_row_pitch %lld != %lld (tight)\n", (long long)block_row_pitch, (long long)tight);
        return false;
    }
    // The library's own trampolines: the slices run as a pipeline -- on the calling thread when one GPU is behind the call (one worker, or a
    // surface that lives on a device), else one pipeline per GPU of the pool with the windows dealt round-robin.
    bc7_enc_settings s7;
    bc6h_enc_settings s6;
    const void* settings = nullptr;
    if (slices > 1 && itwSliceWindow(dxgi_format, source->width, source->height, slice_pixels) > 0 && resolve_trampoline(cmpFunc, dxgi_format, &s7, &s6, &settings)) {
        if (!multithreaded || worker_count() == 1 || is_device_pointer(source->ptr) || is_device_pointer(target))
            return itwCompressImageSlicedEx(source, target, block_row_pitch, dxgi_format, settings, slice_pixels, progress, user);
        itwClearError();
        return sliced_on_pool(source, target, dxgi_format, settings, slice_pixels, slices, progress, user);
    }
    for (int i = 0; i < slices; i++) {
        if (i > 0 && progress && !progress(i, slices, user)) return false;          // allow an early out
        int ylo = (int)((int64_t)i * source->height / slices) & ~0x3;
        int yhi = (int)((int64_t)(i + 1) * source->height / slices) & ~0x3;
        if (yhi > source->height) yhi = source->height;
        if (i == slices - 1 && keeps_partial_blocks(dxgi_format)) yhi = source->height;
        if (yhi > ylo) {
            rgba_surface input = *source;
            input.ptr += (int64_t)input.stride * ylo;
            input.height = yhi - ylo;
            uint8_t* dst = target + block_row_pitch * (ylo >> 2);
            const bool ok = multithreaded ? CompressImageMT(&input, dst, cmpFunc, dxgi_format)
                                          : CompressImageST(&input, dst, cmpFunc, dxgi_format);
            if (!ok) return false;
        }
    }
    return true;
}

rgba_surface itwPadToMultipleOf4(const rgba_surface* input, int pixel_size)
{
    rgba_surface out;
    out.width = (input->width + 3) & ~3;
    out.height = (input->height + 3) & ~3;
    out.stride = out.width * pixel_size;
    out.ptr = (uint8_t*)std::malloc((size_t)out.height * out.stride);
    if (!out.ptr) { std::fprintf(stderr, "itwPadToMultipleOf4: out of memory\n"); std::abort(); }
    for (int y = 0; y < input->height; y++) {
        const uint8_t* rs = input->ptr + (int64_t)y * input->stride;
        uint8_t* rd = out.ptr + (int64_t)y * out.stride;
        std::memcpy(rd, rs, (size_t)input->width * pixel_size);
        for (int x = input->width; x < out.width; x++)                               // trailing pixels
            std::memcpy(rd + (int64_t)x * pixel_size, rs + (int64_t)(input->width - 1) * pixel_size, pixel_size);
    }
    for (int y = input->height; y < out.height; y++) {                              // extra rows
        uint8_t* rd = out.ptr + (int64_t)y * out.stride;
        std::memcpy(rd, rd - out.stride, out.stride);
    }
    return out;
}

void itwFreeSurface(rgba_surface* s)
{
    if (s && s->ptr) { std::free(s->ptr); s->ptr = nullptr; }
}

void itwPadToMultipleOf4Device(const rgba_surface* input, int pixel_size, uint8_t* out_ptr)
{
    const int w = input->width, h = input->height;
    if (w <= 0 || h <= 0) return;
    if ((pixel_size != 4 && pixel_size != 8) || (input->stride & 3) || ((uintptr_t)input->ptr & 3) || ((uintptr_t)out_ptr & 3)) {
        std::fprintf(stderr, "itwPadToMultipleOf4Device: pixel_size must be 4 or 8, pointers and stride 4-byte aligned\n");
        std::abort();
    }
    const int ow = (w + 3) & ~3, oh = (h + 3) & ~3;
    hipStream_t st = (hipStream_t)itwGetStream();
    const dim3 blk(256), grid((unsigned)((ow + 255) / 256), (unsigned)oh);
    if (pixel_size == 4)
        hipLaunchKernelGGL((pad_kernel<1>), grid, blk, 0, st, (const uint32_t*)input->ptr, (int64_t)(input->stride / 4), w, h, (uint32_t*)out_ptr, ow, oh);
    else
        hipLaunchKernelGGL((pad_kernel<2>), grid, blk, 0, st, (const uint32_t*)input->ptr, (int64_t)(input->stride / 4), w, h, (uint32_t*)out_ptr, ow, oh);
    if (hipGetLastError() != hipSuccess) { std::fprintf(stderr, "itwPadToMultipleOf4Device: launch failed\n"); std::abort(); }
}

} // extern "C"
