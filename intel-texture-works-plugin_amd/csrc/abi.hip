// abi.hip -- the extern "C" boundary of libispc_texcomp.so.
//
// Exports exactly the BC symbols of the reference's L1 layer
//   CompressBlocksBC1/BC3/BC6H/BC7   (ispc_texcomp.cpp:417-435)
//   GetProfile_*                     (ispc_texcomp.cpp:20-410)
// plus the itw* extensions of include/itw_amd.h.  The reference forwards these
// calls to ISPC-generated x86 code; here the device boundary (H2D / launch /
// D2H) sits at the same line.  There is no CPU implementation behind this file:
// a HIP failure is reported on stderr and the process aborts, or -- if the host opted in with itwSetErrorMode -- the
// call returns and the message waits in itwLastError() (host_rt.hpp).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/itw_amd.h"
#include "../../include/itw_bc45.h"
#include "../../include/itw_dispatch.h"
#ifdef ITW_TEST_HOOKS
#include "../../include/itw_test_hooks.h"
#endif
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "host_rt.hpp"
#include "kernels.hpp"
#include "x86_math.hpp"

namespace itw {

namespace {
std::atomic<int> g_error_mode{-1};                 // -1 = not decided yet (ITW_ON_ERROR), 0 abort, 1 return
thread_local char t_last_error[384] = {0};
thread_local bool t_has_error = false;

int error_mode() noexcept
{
    int m = g_error_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const char* e = std::getenv("ITW_ON_ERROR");
        m = (e && !std::strcmp(e, "return")) ? 1 : 0;
        g_error_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
}

void fail_hip(const char* what, hipError_t e, const char* file, int line)
{
    (void)hipGetLastError();                        // do not let the sticky error poison the next call
    Failure f;
    std::snprintf(f.msg, sizeof f.msg, "%s failed: %s (%d) at %s:%d", what, hipGetErrorString(e), (int)e, file, line);
    throw f;
}

void fail_msg(const char* fmt, ...)
{
    Failure f;
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(f.msg, sizeof f.msg, fmt, ap);
    va_end(ap);
    throw f;
}

void report_failure(const Failure& f) noexcept
{
    std::snprintf(t_last_error, sizeof t_last_error, "%s", f.msg);
    t_has_error = true;
    if (error_mode() == 0) {
        std::fprintf(stderr, "libispc_texcomp (itw-amd): %s -- no CPU fallback, aborting (itwSetErrorMode(ITW_ON_ERROR_RETURN) "
                             "makes this call return instead)\n", f.msg);
        std::abort();
    }
}

void clear_failure() noexcept { t_has_error = false; t_last_error[0] = 0; }

bool is_device_pointer(const void* p) noexcept
{
    if (!p) return false;
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof(a));
    hipError_t e = hipPointerGetAttributes(&a, p);
    if (e != hipSuccess) { (void)hipGetLastError(); return false; }   // plain malloc'd memory on older runtimes
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

} // namespace itw

namespace {
using itw::is_device_pointer;

// Per host thread: stream for device-resident calls and grow-only staging
// buffers for host-pointer calls.  The reference is called from up to 64 pool
// threads at once (win32Threads.cpp:211-249); thread_local state keeps those
// calls independent without a lock.
struct ThreadCtx {
    hipStream_t user_stream = nullptr;     // itwSetStream
    hipStream_t own_stream  = nullptr;     // staging path: kernels (and the copies of single-chunk calls)
    hipStream_t copy_stream = nullptr;     // staging path: copies of chunked calls, overlapped with own_stream
    hipEvent_t  ev_in[8] = {}, ev_done[8] = {};
    void*  d_in = nullptr;  size_t in_cap = 0;
    void*  d_out = nullptr; size_t out_cap = 0;
    void*  d_ws = nullptr;  size_t ws_cap = 0;      // BC7 inter-family workspace
    hipStream_t ws_stream = nullptr; bool ws_used = false;
    hipEvent_t  ws_event = nullptr;                 // recorded after each BC7 call: orders the workspace across streams
    itw::Bc7Aux aux = {nullptr, nullptr, nullptr, 0, nullptr, false, nullptr, false};   // second stream + fork / join / mid events: the parallel parts of small
                                                    // BC7 calls, the second band of large ones, the second kernel stream of the window pipeline
    itw::Bc7Verdict verdict = {nullptr, nullptr, false, nullptr};   // the pilot's estimate of a staged host-pointer call's first run, left for the host (kernels.hpp)
    bool staged_wide = false;                       // what the last staged BC7 call's estimate said (the shape of this call's first run, until its own is in)
    int    device = -1;
    char   info[256] = {0};
    ~ThreadCtx() {
        // best effort; the runtime may already be gone at process exit
        if (d_in)  (void)hipFree(d_in);
        if (d_out) (void)hipFree(d_out);
        if (d_ws)  (void)hipFree(d_ws);
        if (own_stream) (void)hipStreamDestroy(own_stream);
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        for (auto e : ev_in) if (e) (void)hipEventDestroy(e);
        for (auto e : ev_done) if (e) (void)hipEventDestroy(e);
        if (ws_event) (void)hipEventDestroy(ws_event);
        if (aux.stream) (void)hipStreamDestroy(aux.stream);
        if (aux.fork) (void)hipEventDestroy(aux.fork);
        if (aux.join) (void)hipEventDestroy(aux.join);
        if (aux.mid) (void)hipEventDestroy(aux.mid);
        if (verdict.event) (void)hipEventDestroy(verdict.event);
        if (verdict.host_counts) (void)hipHostFree((void*)verdict.host_counts);
    }
};
thread_local ThreadCtx tls_own;
// A combiner leader (below) works in its DEVICE's shared context instead of its own: whichever of the caller's 8 or 64 pool threads happens to
// lead a burst would otherwise bring its own staging buffers, workspace, streams and pinned words -- tens of milliseconds of hipMalloc the first
// time each thread leads, and 64 copies of the buffers (round 6: BC7 `basic`, one whole-surface CompressImageMT from 64 pool threads: 14.0 -> 4 ms).
thread_local ThreadCtx* tls_lent = nullptr;
#define tls (*(tls_lent ? tls_lent : &tls_own))

// Per-thread resources belong to the device they were created on: a host thread that switches devices (hipSetDevice)
// drops them and starts over.
void bind_thread_to_current_device()
{
    int dev = 0;
    ITW_CHECK(hipGetDevice(&dev));
    if (tls.device == dev) return;
    if (tls.d_in)  { (void)hipFree(tls.d_in);  tls.d_in = nullptr;  tls.in_cap = 0; }
    if (tls.d_out) { (void)hipFree(tls.d_out); tls.d_out = nullptr; tls.out_cap = 0; }
    if (tls.d_ws)  { (void)hipFree(tls.d_ws);  tls.d_ws = nullptr;  tls.ws_cap = 0; tls.ws_used = false; }
    if (tls.own_stream) { (void)hipStreamDestroy(tls.own_stream); tls.own_stream = nullptr; }
    if (tls.copy_stream) { (void)hipStreamDestroy(tls.copy_stream); tls.copy_stream = nullptr; }
    for (auto& e : tls.ev_in) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    for (auto& e : tls.ev_done) if (e) { (void)hipEventDestroy(e); e = nullptr; }
    if (tls.ws_event) { (void)hipEventDestroy(tls.ws_event); tls.ws_event = nullptr; }
    if (tls.aux.stream) { (void)hipStreamDestroy(tls.aux.stream); tls.aux.stream = nullptr; }
    if (tls.aux.fork) { (void)hipEventDestroy(tls.aux.fork); tls.aux.fork = nullptr; }
    if (tls.aux.join) { (void)hipEventDestroy(tls.aux.join); tls.aux.join = nullptr; }
    if (tls.aux.mid) { (void)hipEventDestroy(tls.aux.mid); tls.aux.mid = nullptr; }
    if (tls.verdict.event) { (void)hipEventDestroy(tls.verdict.event); tls.verdict.event = nullptr; }
    if (tls.verdict.host_counts) { (void)hipHostFree((void*)tls.verdict.host_counts); tls.verdict.host_counts = nullptr; tls.verdict.host_counts_dev = nullptr; }
    tls.device = dev;
}

// staging path: streams and events of this thread, created on first use
void ensure_device_ctx()
{
    bind_thread_to_current_device();
    if (!tls.own_stream) ITW_CHECK(hipStreamCreateWithFlags(&tls.own_stream, hipStreamNonBlocking));
    if (!tls.copy_stream) ITW_CHECK(hipStreamCreateWithFlags(&tls.copy_stream, hipStreamNonBlocking));
    for (auto& e : tls.ev_in) if (!e) ITW_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : tls.ev_done) if (!e) ITW_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
}

void* grow(void*& buf, size_t& cap, size_t need)
{
    if (need > cap) {
        if (buf) { void* old = buf; buf = nullptr; cap = 0; ITW_CHECK(hipFree(old)); }
        size_t want = need + need / 4 + 4096;
        void* fresh = nullptr;
        ITW_CHECK(hipMalloc(&fresh, want));
        buf = fresh; cap = want;
    }
    return buf;
}

enum class Fmt { BC1, BC3, BC7, BC6H, BC4, BC5 };

struct Job {
    Fmt fmt;
    const bc7_enc_settings*  s7 = nullptr;
    const bc6h_enc_settings* s6 = nullptr;
};

// The BC7 workspace is per host thread.  Work already queued on another stream may still be using it, so a change
// of stream makes the new stream wait for a library-owned event recorded behind the previous call (never for the
// caller's old stream handle, which may have been destroyed since -- ADVICE r01); calls on one stream are ordered by
// the stream itself.  Growing the workspace frees it first, and hipFree waits for the device.
float* bc7_workspace(int w, int h, hipStream_t st, int64_t wide_max_blocks, const bc7_enc_settings* settings)
{
    bind_thread_to_current_device();
    if (!tls.ws_event) ITW_CHECK(hipEventCreateWithFlags(&tls.ws_event, hipEventDisableTiming));
    if (tls.ws_used && tls.ws_stream != st) ITW_CHECK(hipStreamWaitEvent(st, tls.ws_event, 0));
    float* ws = (float*)grow(tls.d_ws, tls.ws_cap, itw::bc7_workspace_bytes(w, h, wide_max_blocks, settings));
    tls.ws_stream = st; tls.ws_used = true;
    return ws;
}

// Grows the per-thread workspace to `bytes` under the same ordering rules (used to pre-size it for a multi-run call).
void reserve_workspace(size_t bytes, hipStream_t st)
{
    bind_thread_to_current_device();
    if (!tls.ws_event) ITW_CHECK(hipEventCreateWithFlags(&tls.ws_event, hipEventDisableTiming));
    if (tls.ws_used && tls.ws_stream != st) ITW_CHECK(hipStreamWaitEvent(st, tls.ws_event, 0));
    (void)grow(tls.d_ws, tls.ws_cap, bytes);
    tls.ws_stream = st; tls.ws_used = true;
}

// `staged`: a run of a host-pointer call.  Its upload / download overlap the neighbouring runs' kernels.  A BC7 run that is not a band
// (compress() below: round 4's shape, and what content that needs modes 1/3 everywhere still gets) takes the wide shape up to 2^20
// blocks -- scans and single-subset modes side by side on two streams fill the gaps between runs better than a chain of dependent
// launches on one stream (8.75 -> 7.89 ms for a 4096^2 `slow` call in round 2).  Device-resident calls keep the deep shape above
// 262144 blocks (a fifth of the HBM traffic and workspace).
// ITW_STAGED_WIDE_MAX: up to how many blocks a staged run of a host-pointer BC7 call takes the wide launch shape (tuning knob)
int64_t staged_wide_max_blocks()
{
    static const int64_t v = [] { const char* e = std::getenv("ITW_STAGED_WIDE_MAX"); return e ? (int64_t)std::atoll(e) : ((int64_t)1 << 20); }();
    return v < 1 ? 1 : v;
}

// ITW_STAGED_VERDICT_THR: percent of the first staged run's sampled blocks the pilot's estimate may list for the remaining runs to stay deep bands
int staged_verdict_percent()
{
    static const int v = [] { const char* e = std::getenv("ITW_STAGED_VERDICT_THR"); return e ? std::atoi(e) : 80; }();
    return v;
}

void ensure_bc7_aux()
{
    if (tls.aux.stream) return;
    bind_thread_to_current_device();
    ITW_CHECK(hipStreamCreateWithFlags(&tls.aux.stream, hipStreamNonBlocking));
    ITW_CHECK(hipEventCreateWithFlags(&tls.aux.fork, hipEventDisableTiming));
    ITW_CHECK(hipEventCreateWithFlags(&tls.aux.join, hipEventDisableTiming));
    ITW_CHECK(hipEventCreateWithFlags(&tls.aux.mid, hipEventDisableTiming));
    ITW_CHECK(hipEventCreateWithFlags(&tls.verdict.event, hipEventDisableTiming));
    // two pinned, device-visible words the pilot's estimate kernel writes for the host (no copy, no stream: ADVICE r05)
    void* h = nullptr;
    ITW_CHECK(hipHostMalloc(&h, 2 * sizeof(int32_t), hipHostMallocMapped));
    void* d = nullptr;
    ITW_CHECK(hipHostGetDevicePointer(&d, h, 0));
    tls.verdict.host_counts = static_cast<volatile int32_t*>(h);
    tls.verdict.host_counts_dev = static_cast<int32_t*>(d);
}

// `band` >= 0: a staged run of a host-pointer BC7 call that compress() overlaps with its neighbours on two streams; it runs in the deep
// shape on `st` alone, in its own slice of the per-thread workspace (`ws_off`), which compress() has reserved and ordered.
void launch(const Job& j, const uint8_t* d_src, int64_t stride, int w, int h, uint8_t* d_dst, hipStream_t st, bool staged = false, int band = -1,
            size_t ws_off = 0)
{
    switch (j.fmt) {
    case Fmt::BC1:  itw::launch_bc1(d_src, stride, w, h, d_dst, st); break;
    case Fmt::BC3:  itw::launch_bc3(d_src, stride, w, h, d_dst, st); break;
    case Fmt::BC7:
        ensure_bc7_aux();
        tls.aux.wide_max_blocks = staged ? staged_wide_max_blocks() : 0;
        {
            // the per-call fields of the shared aux block go back to their defaults however this call ends (a failure is a C++ exception here,
            // and under ITW_ON_ERROR_RETURN the thread lives on)
            struct Reset { itw::Bc7Aux& a; ~Reset() { a.single = false; a.verdict = nullptr; a.probe = false; } } reset{tls.aux};
            tls.aux.single = band >= 0 || band == -2;
            if (tls.aux.single) {
                // band 0 leaves the pilot's estimate for the host; band -2 is a PROBE: the estimate alone (the run itself takes the wide shape)
                tls.aux.verdict = (band == 0 || band == -2) ? &tls.verdict : nullptr;
                tls.aux.probe = band == -2;
                if (tls.aux.verdict) tls.verdict.valid = false;
                itw::launch_bc7(d_src, stride, w, h, d_dst, *j.s7, reinterpret_cast<float*>(static_cast<uint8_t*>(tls.d_ws) + ws_off), st, &tls.aux);
            } else {
                float* ws = bc7_workspace(w, h, st, tls.aux.wide_max_blocks, j.s7);         // (sized and ordered already when ws_off != 0)
                itw::launch_bc7(d_src, stride, w, h, d_dst, *j.s7, reinterpret_cast<float*>(reinterpret_cast<uint8_t*>(ws) + ws_off), st, &tls.aux);
                ITW_CHECK(hipEventRecord(tls.ws_event, st));
            }
        }
        break;
    case Fmt::BC6H: {
        // the wide shape's workspace shares the per-thread BC7 workspace buffer (same ordering rules)
        const size_t need = itw::bc6h_workspace_bytes(w, h, *j.s6);
        void* ws = nullptr;
        if (need) {
            bind_thread_to_current_device();
            if (!tls.ws_event) ITW_CHECK(hipEventCreateWithFlags(&tls.ws_event, hipEventDisableTiming));
            if (tls.ws_used && tls.ws_stream != st) ITW_CHECK(hipStreamWaitEvent(st, tls.ws_event, 0));
            ws = grow(tls.d_ws, tls.ws_cap, need);
            tls.ws_stream = st; tls.ws_used = true;
        }
        itw::launch_bc6h(d_src, stride, w, h, d_dst, *j.s6, st, ws);
        if (need) ITW_CHECK(hipEventRecord(tls.ws_event, st));
        break;
    }
    case Fmt::BC4:  itw::launch_bc4(d_src, stride, w, h, d_dst, st); break;
    case Fmt::BC5:  itw::launch_bc5(d_src, stride, w, h, d_dst, st); break;
    }
    ITW_CHECK(hipGetLastError());
}

bool coalesce_small_call(const Job& j, const rgba_surface* src, uint8_t* dst, int64_t blocks);
bool heavy_job(const Job& j);
static int64_t bc7_window_min_blocks() { static const int64_t v = [] { const char* e = std::getenv("ITW_BC7_WINDOW_MIN"); return e ? (int64_t)std::atoll(e) : (int64_t)262144; }(); return v; }   // BC7 host-pointer calls from this size take windows (2048^2 `basic` 1.28 -> 1.17 ms, 2896^2 2.46 -> 1.97, `slow` 4.24 -> 3.13)
bool compress_sliced(const Job& j, const rgba_surface* src, uint8_t* dst, int slices, ItwProgressFunc* progress, void* user, int fixed_window = 0,
                     const itw::SlicedPart* share = nullptr, bool leave_verdict = false);

// texel rows of a host surface into the tight staging image
void upload_rows(uint8_t* d_rows, size_t pitch, const uint8_t* hs, int64_t stride, size_t row_bytes, size_t nrows, hipStream_t copy)
{
    if (stride >= (int64_t)row_bytes) {                      // (also for tight rows: measured faster than one linear pageable copy)
        ITW_CHECK(hipMemcpy2DAsync(d_rows, pitch, hs, (size_t)stride, row_bytes, nrows, hipMemcpyHostToDevice, copy));
    } else {
        // bottom-up (negative stride) or overlapping rows: the reference just indexes ptr + y*stride with a
        // signed stride (kernel.ispc:105-151), which a pitched copy cannot express -- stage row by row
        for (size_t y = 0; y < nrows; y++)
            ITW_CHECK(hipMemcpyAsync(d_rows + y * pitch, hs + (int64_t)y * stride, row_bytes, hipMemcpyHostToDevice, copy));
    }
}

void compress(const Job& j, const rgba_surface* src, uint8_t* dst, bool may_coalesce = true)
{
    if (may_coalesce) itw::clear_failure();
    if (!src) itw::fail_msg("null surface");
    const int w = src->width, h = src->height;
    // ISPC formats drop partial blocks (kernel.ispc:600-601); the DirectXTex formats keep them (DirectXTexCompress.cpp:108-116)
    const bool keep_partial = (j.fmt == Fmt::BC4 || j.fmt == Fmt::BC5);
    const int bx = keep_partial ? (w > 0 ? (w + 3) / 4 : 0) : w / 4, by = keep_partial ? (h > 0 ? (h + 3) / 4 : 0) : h / 4;
    if (bx <= 0 || by <= 0) return;                   // nothing to encode: the reference's loops do not run either
    if (!src->ptr || !dst) itw::fail_msg("null texel or destination pointer");
    const int bpb = (j.fmt == Fmt::BC1 || j.fmt == Fmt::BC4) ? 8 : 16;
    const int texel_bytes = (j.fmt == Fmt::BC6H) ? 8 : 4;
    const size_t row_bytes = keep_partial ? (size_t)w * 4 : (size_t)bx * 4 * texel_bytes;
    const size_t rows = keep_partial ? (size_t)h : (size_t)by * 4;
    const size_t out_bytes = (size_t)bx * by * bpb;

    const bool src_dev = is_device_pointer(src->ptr);
    const bool dst_dev = is_device_pointer(dst);

    if (src_dev && dst_dev) {                         // resident pipeline: asynchronous
        launch(j, src->ptr, src->stride, w, h, dst, tls.user_stream);
        return;
    }
    // small host-pointer calls made concurrently by several host threads are joined into one (below)
    if (may_coalesce && !src_dev && !dst_dev && coalesce_small_call(j, src, dst, (int64_t)bx * by)) return;

    // Round 6: a large BC6H call, and a large BC7 call of a profile without an order verdict (everything but `slow`), takes the WINDOW pipeline
    // of the slice loop (compress_sliced below) with no callback: windows of ~131 072 blocks alternate between two kernel streams with the
    // neighbours' copies on the third -- measured against the runs below at 4096^2: BC6H `slow` 4.83 -> 4.3 ms, BC7 `alpha_basic` 3.31 -> 2.9,
    // `alpha_veryfast` 2.46 -> 2.0, `basic` 3.93 -> 3.6 (profiles/r06_host_pointer_path.txt, r06_sliced_timing.jsonl).  The tuning knobs of the runs keep the runs.
    if (!src_dev && !dst_dev && !keep_partial && !std::getenv("ITW_HOST_CHUNKS") && !std::getenv("ITW_HOST_RUNS") && !std::getenv("ITW_HOST_WINDOWS_OFF")) {
        const int64_t blocks = (int64_t)bx * by;
        // A profile WITH an order verdict (`slow`) takes the windows while the last estimate says the bounded order pays (each window runs it on one
        // stream: 5.6 against 6.05 ms for the staged runs below on the bench surface) and the staged runs -- whose remaining runs go wide -- while it
        // says nearly every block needs modes 1/3 (photographs: 7.3 against 7.6 ms as windows).  Either way the call leaves a fresh estimate.
        const bool verdict_profile = j.fmt == Fmt::BC7 && itw::bc7_has_order_verdict(*j.s7);
        const bool bc7w = j.fmt == Fmt::BC7 && blocks >= bc7_window_min_blocks() && itw::bc7_staged_bands_ok() && (!verdict_profile || !tls.staged_wide);
        const bool bc6w = j.fmt == Fmt::BC6H && blocks >= 262144 && j.s6->slow_mode;    // (the other BC6H profiles are PCIe-bound: 2.98 vs 3.02 ms, fewer copies win)
        if (bc7w || bc6w) {
            const int64_t per_window = heavy_job(j) ? 262144 : 131072;
            int64_t windows = (blocks + per_window / 2) / per_window;
            if (windows > by) windows = by;
            if (windows >= 2) { (void)compress_sliced(j, src, dst, (int)windows, nullptr, nullptr, 1, nullptr, bc7w && verdict_profile); return; }
        }
    }

    ensure_device_ctx();
    hipStream_t st = tls.own_stream, cs = tls.copy_stream;
    if (src_dev && tls.user_stream != st) {
        // producer of the device surface may still be running on the caller's stream
        ITW_CHECK(hipStreamSynchronize(tls.user_stream));
    }
    // tight staging pitch, 16-byte aligned rows so the vector load path applies
    const size_t pitch = (row_bytes + 15) & ~(size_t)15;
    uint8_t* in = src_dev ? nullptr : (uint8_t*)grow(tls.d_in, tls.in_cap, pitch * rows);
    const uint8_t* d_src = src_dev ? src->ptr : in;
    const int64_t d_stride = src_dev ? (int64_t)src->stride : (int64_t)pitch;
    uint8_t* d_dst = dst_dev ? dst : (uint8_t*)grow(tls.d_out, tls.out_cap, out_bytes);

    // BC7 / BC6H spend milliseconds per surface: cut the call into runs of block rows so that the upload of run c+1 and
    // the download of run c-1 (copy stream; a pageable copy blocks only this host thread) overlap the kernels of run c.
    // A run must still fill the chip several times over (one block per lane: 1024 workgroups = one round of the BC7
    // scans), or the kernels' tails cost more than the copies save -- measured at 4096^2 (tools/history/misc/host_chunks_probe.py):
    // BC7 slow 9.58 / 9.19 / 9.60 / 11.6 ms for 1 / 2 / 4 / 8 runs, BC6H slow 7.61 / 6.15 / 5.64 / 5.57 ms.
    // BC1/3/4/5 are PCIe-bound: one run.
    // Runs need not be equal: a SHORT FIRST run starts the kernels after an eighth of the upload instead of half of it,
    // and a shorter last run shortens the download nothing can hide (ITW_HOST_RUNS="f0,f1,.." gives the fractions).
    int cut[9] = {0, by, 0, 0, 0, 0, 0, 0, 0};        // run c covers block rows [cut[c], cut[c+1])
    int nch = 1;
    if (!src_dev && (j.fmt == Fmt::BC7 || j.fmt == Fmt::BC6H)) {
        if (j.fmt == Fmt::BC7 && (int64_t)bx * by >= 524288)       { nch = 3; cut[1] = by / 8; cut[2] = by / 8 + by / 2; cut[3] = by; }
        else if (j.fmt == Fmt::BC6H && (int64_t)bx * by >= 262144) { nch = 4; cut[1] = by / 8; cut[2] = by * 3 / 8; cut[3] = by * 11 / 16; cut[4] = by; }
    }
    if (const char* e = std::getenv("ITW_HOST_CHUNKS")) {         // tuning / test knob (1..8 equal runs); 1 disables the overlap
        const int v = std::atoi(e);
        if (v >= 1 && v <= 8 && !src_dev && by >= 4 * v) { nch = v; for (int c = 0; c <= v; c++) cut[c] = (int)((int64_t)by * c / v); }
    }
    if (const char* e = std::getenv("ITW_HOST_RUNS")) {           // tuning knob: cumulative fractions "0.125,0.625" -> 3 runs
        int n = 0; double f[8];
        for (const char* p = e; *p && n < 7;) { char* q = nullptr; f[n] = std::strtod(p, &q); if (q == p) break; n++; p = (*q == ',') ? q + 1 : q; }
        if (n > 0 && !src_dev && by >= 64) {
            nch = n + 1; cut[0] = 0; cut[nch] = by;
            for (int c = 0; c < n; c++) { int r = (int)(by * f[c]); cut[c + 1] = r < cut[c] + 1 ? cut[c] + 1 : (r > by - (n - c) ? by - (n - c) : r); }
        }
    }
    // Round 5: the runs of a BC7 call are BANDS -- run c's kernels go to stream (c + 1) % 2 (the first, short run to the second stream),
    // each run in the deep shape with its own slice of the workspace, so that one run's launch tails are filled by its neighbour's work as
    // soon as that neighbour's texels have arrived (the device-resident path does the same with the two halves of a surface, bc7.hip).
    // Content where nearly every block still needs modes 1/3 (photographs) gains nothing from the bounded order, and for it the WIDE shape,
    // one run after the other, overlaps staged runs better (8.1 against 7.4 ms per 4096^2 call, profiles/history/r05/r05c_*).  Which it is comes
    // from the pilot's estimate (bc7.hip bc7_pilot_estimate): counted behind the first run's {0,2} scan when that run is a band, by a
    // probe on the second stream when it is wide.  The host never waits for it: the first run takes the shape the PREVIOUS call's
    // estimate asked for (successive calls of a save -- mip levels, slices -- hold similar content), later runs this call's as soon as
    // it is in (polled under the uploads), and it is read once more behind the call's last synchronisation for the next call.
    // ITW_STAGED_BANDS=0: round 4's runs, one after the other in the wide shape.
    const bool bands = j.fmt == Fmt::BC7 && nch > 1 && !src_dev && itw::bc7_staged_bands_ok();
    size_t band_off[9] = {0}, probe_off = 0;                 // workspace: [wide runs (they follow each other on st)] [probe] [band 0] [band 1] ...
    if (bands) {
        ensure_bc7_aux();
        size_t total = 0;
        for (int c = 0; c < nch; c++) {
            const int run_rows = (cut[c + 1] - cut[c]) * 4;
            if (run_rows <= 0) continue;
            const size_t b = (itw::bc7_workspace_bytes(w, run_rows, staged_wide_max_blocks(), j.s7) + 255) & ~(size_t)255;
            if (b > total) total = b;
        }
        probe_off = total;
        total += (itw::bc7_workspace_bytes(w, (cut[1] - cut[0]) * 4, 1, j.s7) + 255) & ~(size_t)255;
        for (int c = 0; c < nch; c++) {
            band_off[c] = total;
            const int run_rows = (cut[c + 1] - cut[c]) * 4;
            if (run_rows > 0) total += (itw::bc7_workspace_bytes(w, run_rows, 1, j.s7) + 255) & ~(size_t)255;
        }
        reserve_workspace(total, st);
        ITW_CHECK(hipEventRecord(tls.aux.fork, st));          // the other streams start behind whatever ordered the workspace on the first
        ITW_CHECK(hipStreamWaitEvent(tls.aux.stream, tls.aux.fork, 0));
    } else
    if (j.fmt == Fmt::BC7 || j.fmt == Fmt::BC6H) {
        // Size the workspace once for the most demanding run: growing it frees it, and hipFree waits for the runs in flight.
        // "Most demanding" is not "tallest" (ADVICE r02): a staged BC7 run of up to 2^20 blocks takes the wide shape at ~440
        // B/block while a taller one takes the deep shape at 36 B/block, so the maximum is taken over every run's own need.
        const int64_t wide_max = (!src_dev && !dst_dev) ? staged_wide_max_blocks() : 0;
        size_t need = 0;
        for (int c = 0; c < nch; c++) {
            const int run_rows = (cut[c + 1] - cut[c]) * 4;
            if (run_rows <= 0) continue;
            const size_t b = (j.fmt == Fmt::BC7) ? itw::bc7_workspace_bytes(w, run_rows, wide_max, j.s7) : itw::bc6h_workspace_bytes(w, run_rows, *j.s6);
            if (b > need) need = b;
        }
        if (need) reserve_workspace(need, st);
    }
    // A failure below is a C++ exception (and under ITW_ON_ERROR_RETURN the thread lives on): whatever this call has queued by then on the
    // thread's streams is drained and the workspace's event recorded, so the next call finds d_in / d_out / d_ws free of work in flight.
    struct Unwind {
        bool done; hipStream_t a, b, c;
        ~Unwind() {
            if (done) return;
            (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b);
            if (c) (void)hipStreamSynchronize(c);
            if (tls.ws_event && tls.ws_used) (void)hipEventRecord(tls.ws_event, a);
            (void)hipGetLastError();
        }
    } unwind{false, st, cs, tls.aux.stream};
    hipStream_t copy = (nch > 1) ? cs : st;
    bool shape_wide = bands && tls.staged_wide && itw::bc7_has_order_verdict(*j.s7);   // the shape of the next run (profiles without a verdict: bands)
    bool verdict_pending = false;
    auto poll_verdict = [&](bool wait) {                      // this call's estimate, if it is in (or, `wait`: now that everything is done)
        if (!verdict_pending || !tls.verdict.valid) return;
        if (!wait && hipEventQuery(tls.verdict.event) != hipSuccess) { (void)hipGetLastError(); return; }
        if (wait) ITW_CHECK(hipEventSynchronize(tls.verdict.event));
        // the event has fired: the estimate kernel's last workgroup wrote the two counts into this thread's pinned words before it ended
        const int32_t counts[2] = {tls.verdict.host_counts[0], tls.verdict.host_counts[1]};
        if (counts[1] > 0) tls.staged_wide = (int64_t)counts[0] * 100 > (int64_t)staged_verdict_percent() * counts[1];
        verdict_pending = false;
    };
    int c = 0;
    for (c = 0; c < nch; c++) {
        const int row0 = cut[c], nb = cut[c + 1] - cut[c];
        const size_t y0 = (size_t)row0 * 4;
        const size_t nrows = (rows - y0 < (size_t)nb * 4) ? rows - y0 : (size_t)nb * 4;
        if (bands && c > 0 && verdict_pending) { poll_verdict(false); if (!verdict_pending) shape_wide = tls.staged_wide; }
        hipStream_t run_st = (bands && !shape_wide && ((c + 1) & 1)) ? tls.aux.stream : st;
        if (!src_dev) {
            upload_rows(in + y0 * pitch, pitch, src->ptr + (int64_t)y0 * src->stride, src->stride, row_bytes, nrows, copy);
            if (nch > 1) {
                ITW_CHECK(hipEventRecord(tls.ev_in[c], cs));
                ITW_CHECK(hipStreamWaitEvent(run_st, tls.ev_in[c], 0));
            }
        }
        const uint8_t* run_src = d_src + (int64_t)y0 * d_stride;
        uint8_t* run_dst = d_dst + (size_t)row0 * bx * bpb;
        if (bands && !shape_wide) launch(j, run_src, d_stride, w, (int)nrows, run_dst, run_st, true, c, band_off[c]);
        else                      launch(j, run_src, d_stride, w, (int)nrows, run_dst, run_st, !src_dev && !dst_dev);
        if (bands && c == 0 && shape_wide) {
            // the first run went wide: the estimate comes from a probe on the second stream, behind that run's single-subset modes
            ITW_CHECK(hipStreamWaitEvent(tls.aux.stream, tls.ev_in[0], 0));
            launch(j, run_src, d_stride, w, (int)nrows, run_dst, tls.aux.stream, true, -2, probe_off);
        }
        if (bands && c == 0) verdict_pending = tls.verdict.valid;
        if (!dst_dev && nch > 1) {
            ITW_CHECK(hipEventRecord(tls.ev_done[c], run_st));
            if (c > 0) {                               // download the previous run while this one computes
                const size_t off = (size_t)cut[c - 1] * bx * bpb, len = (size_t)(cut[c] - cut[c - 1]) * bx * bpb;
                ITW_CHECK(hipStreamWaitEvent(cs, tls.ev_done[c - 1], 0));
                ITW_CHECK(hipMemcpyAsync(dst + off, d_dst + off, len, hipMemcpyDeviceToHost, cs));
            }
        }
    }
    if (!dst_dev) {
        if (nch > 1) {
            const size_t off = (size_t)cut[c - 1] * bx * bpb;
            ITW_CHECK(hipStreamWaitEvent(cs, tls.ev_done[c - 1], 0));
            ITW_CHECK(hipMemcpyAsync(dst + off, d_dst + off, out_bytes - off, hipMemcpyDeviceToHost, cs));
            ITW_CHECK(hipStreamSynchronize(cs));
        } else {
            ITW_CHECK(hipMemcpyAsync(dst, d_dst, out_bytes, hipMemcpyDeviceToHost, st));
        }
    }
    if (bands) {                                              // everything back into st; the workspace's event behind all of it
        ITW_CHECK(hipEventRecord(tls.aux.join, tls.aux.stream));
        ITW_CHECK(hipStreamWaitEvent(st, tls.aux.join, 0));
        ITW_CHECK(hipEventRecord(tls.ws_event, st));
    }
    ITW_CHECK(hipStreamSynchronize(st));
    unwind.done = true;
    if (bands) poll_verdict(true);                            // for the next call, if it was not in before
}

// ---- the plugin's slice loop as a pipeline (include/itw_dispatch.h: itwCompressImageSlicedEx) -----------------------------
// IntelPlugin.cpp:851-879 cuts a save into 0x40000-pixel slices and makes one synchronous CompressImageMT/ST call per slice, polling
// SetProgress between them.  Restated literally on a GPU every slice is upload -> a latency-bound launch chain over 1/64 of the
// chip-filling work -> download -> synchronise (BC7 `basic`: 1 274 Mpix/s against 4 175 for one call over the surface).  Here the same
// slices, progress calls and early out run as a PIPELINE: consecutive slices form a WINDOW (the unit of upload / launch / download: about 131 072 blocks, 262 144 for the PCIe-bound formats and the heavy BC7 settings),
// window k's kernels run on stream k % 2 with its
// upload and the previous window's download on the copy stream, and `progress(i, slices)` is called for every slice i of a window -- in
// order, each call only after the slices before it are in `target` -- once that window's bytes have arrived.  A false return stops the
// job: nothing further is issued or copied back, the one window in flight is drained, and every slice below i (and the rest of i's own
// window, at most W - 1 slices more) stays written, like the reference's early out.
std::atomic<int> g_slice_window{0};                // itwSetSliceWindow: 0 = by format and size, -1 = no pipeline (the literal loop)

int slice_window_setting()
{
    int W = g_slice_window.load(std::memory_order_relaxed);
    if (W == 0) {
        static const int env = [] {
            const char* off = std::getenv("ITW_SLICED_PIPELINE");
            if (off && off[0] == '0') return -1;
            const char* e = std::getenv("ITW_SLICE_WINDOW");
            return e ? std::atoi(e) : 0;
        }();
        W = env;
    }
    return W;
}

// `heavy`: BC7 settings that scan every two-subset shape (`slow`, `alpha_slow`: bc7_scans_every_shape) -- twice the work per block and dependent
// launch chains twice as long: windows twice as large (4096^2 `slow`, ms per sliced call with 8 / 16 / 32 slices per
// window: bench surface 5.91 / 5.57 / 5.86, photograph 8.18 / 7.62 / 7.73)
int slice_window(Fmt fmt, int64_t slice_blocks, int slices, bool heavy)
{
    int W = slice_window_setting();
    if (W <= 0) {
        // BC7 / BC6H: a window must fill the chip (131 072 blocks = one block per lane of 2 waves per SIMD; up to three windows are in flight);
        // BC1 / BC3 / BC4 / BC5 are PCIe-bound: fewer, larger copies
        const bool compute = fmt == Fmt::BC7 || fmt == Fmt::BC6H;
        const int64_t target = (compute && !heavy) ? 131072 : 262144;
        const int64_t per = slice_blocks < 1 ? 1 : slice_blocks;
        W = (int)((target + per / 2) / per);
        // No cap by slice count (the first version kept >= 8 windows per job): a window is 0.3-1 ms of work, so a job long enough to show a
        // progress bar has many; and windows SMALLER than the target run the deep launch shape on a fraction of the chip -- 2048^2 `basic` with 8
        // windows of 32 768 blocks 1.79 ms, with 2 of 131 072 1.13 (one call 1.16); 1024^2 with 4 windows 1.12 ms, as one 0.48 (literal loop 0.85).
    }
    if (W < 1) W = 1;
    if (W > slices) W = slices;
    return W;
}
bool heavy_job(const Job& j) { return j.fmt == Fmt::BC7 && j.s7 && itw::bc7_scans_every_shape(*j.s7); }

struct SliceRows { int64_t y0, y1; };
// rows of slice i of `slices` (IntelPlugin.cpp:861-865); the formats that keep partial blocks end the last slice at `height`
SliceRows slice_rows(int i, int slices, int height, bool keep_partial)
{
    SliceRows r;
    r.y0 = ((int64_t)i * height / slices) & ~(int64_t)3;
    r.y1 = ((int64_t)(i + 1) * height / slices) & ~(int64_t)3;
    if (r.y1 > height) r.y1 = height;
    if (i == slices - 1 && keep_partial) r.y1 = height;
    return r;
}

// returns true when every slice was encoded, false when `progress` stopped the job
// `fixed_window` > 0 fixes W (compress() passes 1: its "slices" are the windows).
// `share`: this thread runs only the windows share->part, + share->parts, ... (one pipeline per GPU of the pool, dispatch.hip) and reports them
// through share->retired instead of polling `progress`.
// `leave_verdict` (BC7 profiles with an order verdict, whole-surface host-pointer calls): the first window also counts the pilot's estimate, and the
// call ends by reading it into tls.staged_wide -- what the NEXT such call's shape goes by (compress()).
bool compress_sliced(const Job& j, const rgba_surface* src, uint8_t* dst, int slices, ItwProgressFunc* progress, void* user, int fixed_window,
                     const itw::SlicedPart* share, bool leave_verdict)
{
    if (!src) itw::fail_msg("null surface");
    const int w = src->width, h = src->height;
    const bool keep_partial = (j.fmt == Fmt::BC4 || j.fmt == Fmt::BC5);
    const int bx = keep_partial ? (w > 0 ? (w + 3) / 4 : 0) : w / 4, by = keep_partial ? (h > 0 ? (h + 3) / 4 : 0) : h / 4;
    if (slices < 1) slices = 1;
    auto poll = [&](int first, int last) {               // progress(i) for i in [first, last], i < slices; false = stop
        for (int i = first; i <= last && i < slices; i++)
            if (i > 0 && progress && !progress(i, slices, user)) return false;
        return true;
    };
    if (bx <= 0 || by <= 0) return share ? true : poll(1, slices - 1);   // nothing to encode: the reference's loop still polls
    if (!src->ptr || !dst) itw::fail_msg("null texel or destination pointer");
    const int bpb = (j.fmt == Fmt::BC1 || j.fmt == Fmt::BC4) ? 8 : 16;
    const int texel_bytes = (j.fmt == Fmt::BC6H) ? 8 : 4;
    const size_t row_bytes = keep_partial ? (size_t)w * 4 : (size_t)bx * 4 * texel_bytes;
    const size_t rows = keep_partial ? (size_t)h : (size_t)by * 4;
    const size_t out_bytes = (size_t)bx * by * bpb;
    const bool src_dev = is_device_pointer(src->ptr), dst_dev = is_device_pointer(dst);

    const int W = fixed_window > 0 ? (fixed_window < slices ? fixed_window : slices) : slice_window(j.fmt, (int64_t)bx * by / slices, slices, heavy_job(j));
    const int nwin = (slices + W - 1) / W;
    const int part = share ? share->part : 0, parts = share ? (share->parts < 1 ? 1 : share->parts) : 1;
    const int nlocal = part < nwin ? (nwin - part + parts - 1) / parts : 0;       // this thread's windows: part, part + parts, ...
    if (nlocal == 0) return true;
    if (nwin == 1 && !share && fixed_window == 0) {
        // one window = the whole surface: the ordinary call (which picks its launch shape by size: the wide one for small surfaces), then the polls
        compress(j, src, dst, false);
        if (src_dev && dst_dev) ITW_CHECK(hipStreamSynchronize(tls.user_stream));
        return poll(1, slices - 1);
    }

    ensure_device_ctx();
    ensure_bc7_aux();
    hipStream_t k0 = tls.own_stream, k1 = tls.aux.stream, cs = tls.copy_stream;
    // whatever produced a device surface (or still reads the destination) on the caller's stream comes first
    if ((src_dev || dst_dev) && tls.user_stream != k0) ITW_CHECK(hipStreamSynchronize(tls.user_stream));
    const size_t pitch = (row_bytes + 15) & ~(size_t)15;
    uint8_t* in = src_dev ? nullptr : (uint8_t*)grow(tls.d_in, tls.in_cap, pitch * rows);
    const uint8_t* d_src = src_dev ? src->ptr : in;
    const int64_t d_stride = src_dev ? (int64_t)src->stride : (int64_t)pitch;
    uint8_t* d_dst = dst_dev ? dst : (uint8_t*)grow(tls.d_out, tls.out_cap, out_bytes);

    // BC7: one slice of the per-thread workspace per kernel stream, sized for the tallest window (deep shape: `single` in launch())
    size_t ws_off[2] = {0, 0};
    if (j.fmt == Fmt::BC7) {
        int64_t tallest = 0;
        for (int k = 0; k < nwin; k++) {
            const int s1 = (k + 1) * W < slices ? (k + 1) * W : slices;
            const int64_t r = slice_rows(s1 - 1, slices, h, false).y1 - slice_rows(k * W, slices, h, false).y0;
            if (r > tallest) tallest = r;
        }
        const size_t b = (itw::bc7_workspace_bytes(w, (int)tallest, 1, j.s7) + 255) & ~(size_t)255;
        ws_off[1] = b;
        reserve_workspace(2 * b, k0);
    }
    ITW_CHECK(hipEventRecord(tls.aux.fork, k0));          // the second stream starts behind whatever ordered the workspace on the first
    ITW_CHECK(hipStreamWaitEvent(k1, tls.aux.fork, 0));

    // However this ends, nothing of this call is left in flight on the thread's streams and buffers (a failure is a C++ exception here)
    struct Drain {
        hipStream_t a, b, c; bool bc7;
        ~Drain() {
            (void)hipStreamSynchronize(a); (void)hipStreamSynchronize(b); (void)hipStreamSynchronize(c);
            if (bc7 && tls.ws_event) (void)hipEventRecord(tls.ws_event, a);
            (void)hipGetLastError();
        }
    } drain{k0, k1, cs, j.fmt == Fmt::BC7};

    struct Window { int s0, s1; int64_t y0, y1; int row0, nb; };
    auto window = [&](int k) {
        Window v;
        v.s0 = k * W; v.s1 = (k + 1) * W < slices ? (k + 1) * W : slices;
        v.y0 = slice_rows(v.s0, slices, h, keep_partial).y0;
        v.y1 = slice_rows(v.s1 - 1, slices, h, keep_partial).y1;
        v.row0 = (int)(v.y0 / 4);
        v.nb = (int)((v.y1 - v.y0 + 3) / 4);
        return v;
    };
    // (issue / retire take the LOCAL index i of this thread's window part + i * parts: streams, workspace slices and events alternate by it)
    auto issue = [&](int i) {
        const int k = i;
        const Window v = window(part + i * parts);
        if (v.y1 <= v.y0) return;
        hipStream_t ks = (k & 1) ? k1 : k0;
        const size_t nrows = (size_t)(v.y1 - v.y0);
        if (!src_dev) {
            upload_rows(in + (size_t)v.y0 * pitch, pitch, src->ptr + v.y0 * (int64_t)src->stride, src->stride, row_bytes, nrows, cs);
            ITW_CHECK(hipEventRecord(tls.ev_in[k & 7], cs));
            ITW_CHECK(hipStreamWaitEvent(ks, tls.ev_in[k & 7], 0));
        }
        launch(j, d_src + v.y0 * d_stride, d_stride, w, (int)nrows, d_dst + (size_t)v.row0 * bx * bpb, ks, true, j.fmt == Fmt::BC7 ? ((leave_verdict && i == 0) ? 0 : 1) : -1,
               ws_off[k & 1]);
        ITW_CHECK(hipEventRecord(tls.ev_done[k & 7], ks));
    };
    auto retire = [&](int i) {                            // the window's bytes into `dst`; returns when they are there
        const int k = i;
        const Window v = window(part + i * parts);
        if (v.y1 <= v.y0) return;
        if (dst_dev) { ITW_CHECK(hipEventSynchronize(tls.ev_done[k & 7])); return; }
        const size_t off = (size_t)v.row0 * bx * bpb, len = (size_t)v.nb * bx * bpb;
        ITW_CHECK(hipStreamWaitEvent(cs, tls.ev_done[k & 7], 0));
        ITW_CHECK(hipMemcpyAsync(dst + off, d_dst + off, len, hipMemcpyDeviceToHost, cs));
        ITW_CHECK(hipStreamSynchronize(cs));
    };
    // Two windows AHEAD of the one being retired are issued: when window k-1's kernels end, window k is running on the other stream and window
    // k+1 -- same stream as k-1, behind it in stream order -- already has its texels on the device.  (With one window of lookahead the upload of
    // k+1 only started after k-1 had been retired, and for its 0.16 ms the chip ran window k alone.  4096^2, 64 slices, ms per call with a lookahead of
    // 1 / 2 / 3: `basic` 3.81 / 3.58 / 3.59, `alpha_basic` 3.20 / 2.86 / 2.94, `slow` 6.50 / 5.96 / 5.96, BC6H `slow` 4.35 / 4.29 / 4.27, BC1 1.48 / 1.50 / 1.49.)
    static const int depth = [] { const char* e = std::getenv("ITW_SLICE_LOOKAHEAD"); const int v = e ? std::atoi(e) : 2; return v < 1 ? 1 : (v > 4 ? 4 : v); }();
    auto stopped = [&] { return share && share->stop && share->stop->load(std::memory_order_acquire); };
    for (int i = 0; i < depth && i < nlocal; i++) issue(i);
    for (int i = 0; i < nlocal; i++) {
        if (stopped()) return false;
        if (i + depth < nlocal) issue(i + depth);
        retire(i);
        const Window v = window(part + i * parts);
        if (share) { if (share->retired) share->retired(v.s0, v.s1, share->ctx); }
        else if (!poll(v.s0 + 1, v.s1)) return false;    // the windows behind it are in flight: ~Drain waits for them, their bytes are not copied back
    }
    if (leave_verdict && tls.verdict.valid) {            // the first window's estimate, for the next call (its kernel is long done: every window has been retired)
        ITW_CHECK(hipEventSynchronize(tls.verdict.event));
        const int32_t listed = tls.verdict.host_counts[0], sampled = tls.verdict.host_counts[1];
        if (sampled > 0) tls.staged_wide = (int64_t)listed * 100 > (int64_t)staged_verdict_percent() * sampled;
    }
    return true;
}

// format code + settings pointer of the dispatch layer -> Job (the settings struct must outlive the job)
Job job_of(int dxgi_format, const void* settings)
{
    Job j;
    switch (dxgi_format) {
    case ITW_DXGI_FORMAT_BC1_UNORM: case ITW_DXGI_FORMAT_BC1_UNORM_SRGB: j.fmt = Fmt::BC1; break;
    case ITW_DXGI_FORMAT_BC3_UNORM: case ITW_DXGI_FORMAT_BC3_UNORM_SRGB: j.fmt = Fmt::BC3; break;
    case ITW_DXGI_FORMAT_BC4_UNORM: j.fmt = Fmt::BC4; break;
    case ITW_DXGI_FORMAT_BC5_UNORM: j.fmt = Fmt::BC5; break;
    case ITW_DXGI_FORMAT_BC6H_UF16: case ITW_DXGI_FORMAT_BC6H_SF16: j.fmt = Fmt::BC6H; j.s6 = static_cast<const bc6h_enc_settings*>(settings); break;
    case ITW_DXGI_FORMAT_BC7_UNORM: case ITW_DXGI_FORMAT_BC7_UNORM_SRGB: j.fmt = Fmt::BC7; j.s7 = static_cast<const bc7_enc_settings*>(settings); break;
    default: itw::fail_msg("DXGI format %d is not one this library encodes", dxgi_format);
    }
    if ((j.fmt == Fmt::BC7 || j.fmt == Fmt::BC6H) && !settings) itw::fail_msg("null settings for a BC7 / BC6H job");
    return j;
}

int slices_of(const rgba_surface* s, int64_t slice_pixels)
{
    if (slice_pixels <= 0) slice_pixels = 0x40000;                               // IntelPlugin.cpp:851
    int64_t slices = ((int64_t)s->width * s->height) / slice_pixels;
    if (slices < 1) slices = 1;
    if (slices > (1 << 24)) itw::fail_msg("%lld slices", (long long)slices);
    return (int)slices;
}

// ---- joining concurrent small calls ----------------------------------------------------------------------------------
// The reference's dispatch layer cuts every 0x40000-pixel slice into one band per pool thread and calls the ABI from all
// of them at once (win32Threads.cpp:211-249: 64 threads -> 64 calls of 8 texel rows at 4096 wide).  Each such call alone
// is a few workgroups plus two PCIe copies and a synchronisation -- 4096 of them per 4096^2 image.  Concurrent small
// host-pointer calls are therefore combined ("flat combining"): a caller queues its request; whoever finds no leader
// becomes the leader, takes everything queued for its device, merges requests that continue each other in memory (same
// format, settings, width and stride; src and dst of the next band start where the previous one ends -- exactly what
// win32Threads.cpp:223-230 produces), runs the merged surfaces as ordinary calls on its own stream and staging buffers,
// marks the requests done and hands leadership on.  A single-threaded caller always finds the queue empty and pays one
// uncontended mutex.  ITW_COALESCE=0 disables it.
struct Pending {
    Job job;
    bc7_enc_settings s7;
    bc6h_enc_settings s6;
    rgba_surface src;
    uint8_t* dst = nullptr;
    bool done = false, failed = false;
    char msg[384] = {0};
};

// ITW_COALESCE_DEBUG=1: per device, at process exit: bursts, batches (leader rounds), requests, merged calls, microseconds leaders waited
struct CombinerStats { std::atomic<long long> bursts{0}, batches{0}, requests{0}, calls{0}, wait_us{0}; };
CombinerStats g_cstats;
bool combiner_debug()
{
    static const bool on = [] {
        const char* e = std::getenv("ITW_COALESCE_DEBUG");
        const bool v = e && e[0] == '1';
        if (v) std::atexit([] {
            std::fprintf(stderr, "itw combiner: %lld bursts, %lld batches, %lld requests, %lld merged calls, leaders waited %lld us\n", g_cstats.bursts.load(),
                         g_cstats.batches.load(), g_cstats.requests.load(), g_cstats.calls.load(), g_cstats.wait_us.load());
        });
        return v;
    }();
    return on;
}

struct Combiner {
    std::mutex m;
    std::condition_variable cv;
    std::vector<Pending*> queue;
    bool leader = false;
    ThreadCtx* ctx = nullptr;          // the device's shared context: used by whoever leads (one thread at a time), created by the first leader, never freed
    int burst = 0;                     // requests served since the queue was last idle
    int expected = 1;                  // size of the previous burst: the reference's pool submits the same number of bands
                                       // for every slice (win32Threads.cpp:217-231), so the next burst will be this large too
};
constexpr int kMaxDevices = 64;
Combiner g_combiner[kMaxDevices];
// Calls up to this size are combined.  65 536 until round 6 ("larger calls fill the chip on their own") -- but eight pool threads handing over a 4096^2
// surface make eight concurrent calls of 131 072 blocks, each staging through its own thread's buffers and streams: BC7 `slow` 7.8 ms against 5.8 for one
// call, `basic` 4.06 against 3.72 (profiles/r06_reference_caller_timing.jsonl before the change).  Merged they are one call, which takes the windows.
constexpr int64_t kCoalesceMaxBlocks = 524288;       // (two pool threads halve a 4096^2 surface into calls of this size)

bool coalescing_enabled()
{
    static const bool on = [] { const char* e = std::getenv("ITW_COALESCE"); return !(e && e[0] == '0'); }();
    return on;
}

bool same_settings(const Pending& a, const Pending& b)
{
    if (a.job.fmt != b.job.fmt) return false;
    if (a.job.fmt == Fmt::BC7) {
        // fields only: padding differs between stack objects, and refineIterations[7] is uninitialised storage in the RGB
        // presets (ispc_texcomp.cpp:20-189 never write it) -- it matters only when mode 7 runs (kernel.ispc:1419)
        const bc7_enc_settings &x = a.s7, &y = b.s7;
        for (int i = 0; i < 4; i++) if (x.mode_selection[i] != y.mode_selection[i]) return false;
        for (int i = 0; i < 7; i++) if (x.refineIterations[i] != y.refineIterations[i]) return false;
        const bool mode7 = x.mode_selection[1] && x.fastSkipTreshold_mode7 > 0;
        if (mode7 && x.refineIterations[7] != y.refineIterations[7]) return false;
        return x.skip_mode2 == y.skip_mode2 && x.fastSkipTreshold_mode1 == y.fastSkipTreshold_mode1 &&
               x.fastSkipTreshold_mode3 == y.fastSkipTreshold_mode3 && x.fastSkipTreshold_mode7 == y.fastSkipTreshold_mode7 &&
               x.mode45_channel0 == y.mode45_channel0 && x.refineIterations_channel == y.refineIterations_channel && x.channels == y.channels;
    }
    if (a.job.fmt == Fmt::BC6H) {
        const bc6h_enc_settings &x = a.s6, &y = b.s6;
        return x.slow_mode == y.slow_mode && x.fast_mode == y.fast_mode && x.refineIterations_1p == y.refineIterations_1p &&
               x.refineIterations_2p == y.refineIterations_2p && x.fastSkipTreshold == y.fastSkipTreshold;
    }
    return true;
}

// b continues a: next rows of the same image, next block rows of the same stream
bool continues(const Pending& a, const Pending& b)
{
    if (!same_settings(a, b) || a.src.width != b.src.width || a.src.stride != b.src.stride) return false;
    if ((a.src.height & 3) || a.src.stride <= 0) return false;
    const int bpb = (a.job.fmt == Fmt::BC1 || a.job.fmt == Fmt::BC4) ? 8 : 16;
    const bool keep = a.job.fmt == Fmt::BC4 || a.job.fmt == Fmt::BC5;
    const int64_t bx = keep ? (a.src.width + 3) / 4 : a.src.width / 4;
    return b.src.ptr == a.src.ptr + (int64_t)a.src.height * a.src.stride && b.dst == a.dst + (int64_t)(a.src.height / 4) * bx * bpb;
}

void run_batch(std::vector<Pending*>& batch)
{
    std::sort(batch.begin(), batch.end(), [](const Pending* a, const Pending* b) { return a->src.ptr < b->src.ptr; });
    size_t i = 0;
    while (i < batch.size()) {
        size_t k = i + 1;
        rgba_surface merged = batch[i]->src;
        while (k < batch.size() && continues(*batch[k - 1], *batch[k]) && (int64_t)merged.height + batch[k]->src.height < (1 << 30)) {
            merged.height += batch[k]->src.height;
            k++;
        }
        Job j = batch[i]->job;
        j.s7 = &batch[i]->s7; j.s6 = &batch[i]->s6;
        if (combiner_debug()) g_cstats.calls++;
        try { compress(j, &merged, batch[i]->dst, false); }
        catch (const itw::Failure& f) {
            for (size_t t = i; t < k; t++) { batch[t]->failed = true; std::snprintf(batch[t]->msg, sizeof batch[t]->msg, "%s", f.msg); }
        }
        i = k;
    }
}

bool coalesce_small_call(const Job& j, const rgba_surface* src, uint8_t* dst, int64_t blocks)
{
    if (blocks > kCoalesceMaxBlocks || !coalescing_enabled()) return false;
    bind_thread_to_current_device();
    if (tls.device < 0 || tls.device >= kMaxDevices) return false;
    Combiner& c = g_combiner[tls.device];
    Pending me;
    me.job = j;
    if (j.s7) me.s7 = *j.s7;
    if (j.s6) me.s6 = *j.s6;
    me.src = *src;
    me.dst = dst;

    std::unique_lock<std::mutex> lk(c.m);
    c.queue.push_back(&me);
    while (!me.done) {
        if (c.leader) { c.cv.wait(lk); continue; }
        c.leader = true;                                  // lead one batch (it contains this thread's request), then hand on
        if (c.burst == 0 && c.expected > 1) {
            // start of a burst from a pool of threads released by one event: they arrive over some tens of microseconds.
            // Wait for as many requests as the previous burst had, as long as the queue keeps growing, so that the whole
            // slice becomes ONE merged call instead of a first small batch and a second one with the stragglers.
            const auto t0 = std::chrono::steady_clock::now();
            auto grown = t0;
            size_t seen = c.queue.size();
            while ((int)c.queue.size() < c.expected) {
                lk.unlock();
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(8);
                while (std::chrono::steady_clock::now() < until) std::this_thread::yield();
                lk.lock();
                const auto now = std::chrono::steady_clock::now();
                if (c.queue.size() != seen) { seen = c.queue.size(); grown = now; }
                if (now - grown > std::chrono::microseconds(80) || now - t0 > std::chrono::microseconds(600)) break;
            }
            if (combiner_debug()) g_cstats.wait_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
        }
        std::vector<Pending*> batch;
        batch.swap(c.queue);
        if (combiner_debug()) { g_cstats.batches++; g_cstats.requests += (long long)batch.size(); if (c.burst == 0) g_cstats.bursts++; }
        c.burst += (int)batch.size();
        lk.unlock();
        if (!c.ctx) c.ctx = new ThreadCtx;                // (only the leader touches c.ctx, and there is one leader at a time)
        tls_lent = c.ctx;
        try { run_batch(batch); }
        catch (...) {                                     // anything but an itw::Failure (those are caught per merged call):
            for (Pending* p : batch)                      // the requests must still be released and leadership handed on
                if (!p->failed) { p->failed = true; std::snprintf(p->msg, sizeof p->msg, "unexpected C++ exception while running a combined call"); }
        }
        tls_lent = nullptr;
        lk.lock();
        for (Pending* p : batch) p->done = true;
        if (c.queue.empty()) { c.expected = c.burst; c.burst = 0; }      // burst over
        c.leader = false;
        c.cv.notify_all();
    }
    lk.unlock();
    if (me.failed) { itw::Failure f; std::snprintf(f.msg, sizeof f.msg, "%s", me.msg); throw f; }
    return true;
}

// ---- quality presets (values: ispc_texcomp.cpp:20-410) ----------------------
struct Bc7Preset {
    int channels;
    bool sel[4];            // modes {0,2} {1,3,7} {4,5} {6}
    bool skip_mode2;
    int n1, n3, n7;         // partitions tried after PCA ranking
    int ch0, refine_channel;
    int refine[8];          // per mode; [7] < 0 = "left untouched" (the RGB profiles never write it)
};

constexpr Bc7Preset kUltrafast      {3, {false, false, false, true}, true,  3,  1,  0, 0, 0, {2, 2, 2, 1, 2, 2, 1, -1}};
constexpr Bc7Preset kVeryfast       {3, {false, true,  false, true}, true,  3,  1,  0, 0, 0, {2, 2, 2, 1, 2, 2, 1, -1}};
constexpr Bc7Preset kFast           {3, {false, true,  false, true}, true,  12, 4,  0, 0, 0, {2, 2, 2, 1, 2, 2, 2, -1}};
constexpr Bc7Preset kBasic          {3, {true,  true,  true,  true}, true,  12, 8,  0, 0, 2, {2, 2, 2, 2, 2, 2, 2, -1}};
constexpr Bc7Preset kSlow           {3, {true,  true,  true,  true}, false, 64, 64, 0, 0, 4, {4, 4, 4, 4, 4, 4, 4, -1}};
constexpr Bc7Preset kAlphaUltrafast {4, {false, false, true,  true}, true,  0,  0,  4, 3, 1, {2, 1, 2, 1, 1, 1, 2, 2}};
constexpr Bc7Preset kAlphaVeryfast  {4, {false, true,  true,  true}, true,  0,  0,  4, 3, 2, {2, 1, 2, 1, 2, 2, 2, 2}};
constexpr Bc7Preset kAlphaFast      {4, {false, true,  true,  true}, true,  4,  4,  8, 3, 2, {2, 1, 2, 1, 2, 2, 2, 2}};
constexpr Bc7Preset kAlphaBasic     {4, {true,  true,  true,  true}, true,  12, 8,  8, 0, 2, {2, 2, 2, 2, 2, 2, 2, 2}};
constexpr Bc7Preset kAlphaSlow      {4, {true,  true,  true,  true}, false, 64, 64, 64, 0, 4, {4, 4, 4, 4, 4, 4, 4, 4}};

void apply(const Bc7Preset& p, bc7_enc_settings* s)
{
    // field-wise on purpose: padding bytes and (for RGB presets) refineIterations[7]
    // keep whatever the caller had there, like the reference's assignment lists.
    s->channels = p.channels;
    for (int i = 0; i < 4; i++) s->mode_selection[i] = p.sel[i];
    s->skip_mode2 = p.skip_mode2;
    s->fastSkipTreshold_mode1 = p.n1;
    s->fastSkipTreshold_mode3 = p.n3;
    s->fastSkipTreshold_mode7 = p.n7;
    s->mode45_channel0 = p.ch0;
    s->refineIterations_channel = p.refine_channel;
    for (int i = 0; i < 8; i++) if (p.refine[i] >= 0) s->refineIterations[i] = p.refine[i];
}

struct Bc6hPreset { bool slow, fast; int n, r1p, r2p; };
void apply(const Bc6hPreset& p, bc6h_enc_settings* s)
{
    s->slow_mode = p.slow; s->fast_mode = p.fast;
    s->fastSkipTreshold = p.n; s->refineIterations_1p = p.r1p; s->refineIterations_2p = p.r2p;
}

// ---- device self-test kernels (libispc_texcomp_test.so only) ------------------
#ifdef ITW_TEST_HOOKS
__global__ void k_test_rcp(const float* in, float* out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = itw::ispc_rcp(in[i], itw::global_seed_tables());
}
// Both implementations on every element: the general model on the global tables and the LDS fast path the BC7
// kernels use.  Where they disagree the output is corrupted on purpose so the oracle comparison fails.
__global__ void k_test_rsqrt(const float* in, float* out, int64_t n)
{
    __shared__ unsigned short s16[2048];
    __shared__ uint32_t s32[2048];
    const itw::SeedTables T = itw::stage_seed_tables_fast(s16, s32, threadIdx.x, blockDim.x);
    __syncthreads();
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float g = itw::ispc_rsqrt<false>(in[i], itw::global_seed_tables());
    const float f = itw::ispc_rsqrt<true>(in[i], T);
    const uint32_t gb = __float_as_uint(g), fb = __float_as_uint(f);
    out[i] = (gb == fb) ? g : ((g != g) ? 0.0f : __uint_as_float(gb ^ 1u));
}
__global__ void k_test_f2i(const float* in, int32_t* out, int64_t n)
{
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = itw::f2i_x86(in[i]);
}
#endif // ITW_TEST_HOOKS

} // namespace

namespace itw {

bool sliced_part(const rgba_surface* source, uint8_t* target, int dxgi_format, const void* settings, int64_t slice_pixels, const SlicedPart& part)
{
    bool done = false;
    clear_failure();
    const bool ok = guarded([&] {
        if (!source) fail_msg("null surface");
        const Job j = job_of(dxgi_format, settings);
        done = compress_sliced(j, source, target, slices_of(source, slice_pixels), nullptr, nullptr, 0, &part);
    });
    return ok && done;
}

int sliced_windows(int dxgi_format, const void* settings, int width, int height, int64_t slice_pixels, int* window_slices)
{
    const int W = itwSliceWindowFor(dxgi_format, settings, width, height, slice_pixels);
    if (window_slices) *window_slices = W;
    if (W <= 0) return 0;
    if (slice_pixels <= 0) slice_pixels = 0x40000;
    int64_t slices = ((int64_t)width * height) / slice_pixels;
    if (slices < 1) slices = 1;
    return (int)((slices + W - 1) / W);
}

} // namespace itw

extern "C" {

void GetProfile_ultrafast(bc7_enc_settings* s) { apply(kUltrafast, s); }
void GetProfile_veryfast (bc7_enc_settings* s) { apply(kVeryfast, s); }
void GetProfile_fast     (bc7_enc_settings* s) { apply(kFast, s); }
void GetProfile_basic    (bc7_enc_settings* s) { apply(kBasic, s); }
void GetProfile_slow     (bc7_enc_settings* s) { apply(kSlow, s); }
void GetProfile_alpha_ultrafast(bc7_enc_settings* s) { apply(kAlphaUltrafast, s); }
void GetProfile_alpha_veryfast (bc7_enc_settings* s) { apply(kAlphaVeryfast, s); }
void GetProfile_alpha_fast     (bc7_enc_settings* s) { apply(kAlphaFast, s); }
void GetProfile_alpha_basic    (bc7_enc_settings* s) { apply(kAlphaBasic, s); }
void GetProfile_alpha_slow     (bc7_enc_settings* s) { apply(kAlphaSlow, s); }

void GetProfile_bc6h_veryfast(bc6h_enc_settings* s) { apply(Bc6hPreset{false, true,  0,  0, 0}, s); }
void GetProfile_bc6h_fast    (bc6h_enc_settings* s) { apply(Bc6hPreset{false, true,  2,  0, 1}, s); }
void GetProfile_bc6h_basic   (bc6h_enc_settings* s) { apply(Bc6hPreset{false, false, 4,  2, 2}, s); }
void GetProfile_bc6h_slow    (bc6h_enc_settings* s) { apply(Bc6hPreset{true,  false, 10, 2, 2}, s); }
void GetProfile_bc6h_veryslow(bc6h_enc_settings* s) { apply(Bc6hPreset{true,  false, 32, 2, 2}, s); }

void CompressBlocksBC1(const rgba_surface* src, uint8_t* dst)
{
    itw::guarded([&] { Job j; j.fmt = Fmt::BC1; compress(j, src, dst); });
}
void CompressBlocksBC3(const rgba_surface* src, uint8_t* dst)
{
    itw::guarded([&] { Job j; j.fmt = Fmt::BC3; compress(j, src, dst); });
}
void CompressBlocksBC7(const rgba_surface* src, uint8_t* dst, bc7_enc_settings* settings)
{
    itw::guarded([&] {
        if (!settings) itw::fail_msg("null bc7 settings");
        Job j; j.fmt = Fmt::BC7; j.s7 = settings; compress(j, src, dst);
    });
}
void CompressBlocksBC6H(const rgba_surface* src, uint8_t* dst, bc6h_enc_settings* settings)
{
    itw::guarded([&] {
        if (!settings) itw::fail_msg("null bc6h settings");
        Job j; j.fmt = Fmt::BC6H; j.s6 = settings; compress(j, src, dst);
    });
}

void CompressBlocksBC4(const rgba_surface* src, uint8_t* dst)
{
    itw::guarded([&] { Job j; j.fmt = Fmt::BC4; compress(j, src, dst); });
}
void CompressBlocksBC5(const rgba_surface* src, uint8_t* dst)
{
    itw::guarded([&] { Job j; j.fmt = Fmt::BC5; compress(j, src, dst); });
}

bool itwCompressImageSlicedEx(const rgba_surface* source, uint8_t* target, int64_t block_row_pitch, int dxgi_format, const void* settings,
                              int64_t slice_pixels, ItwProgressFunc* progress, void* user)
{
    bool done = false;
    itw::clear_failure();
    const bool ok = itw::guarded([&] {
        if (!source) itw::fail_msg("null surface");
        const Job j = job_of(dxgi_format, settings);
        const bool keep = j.fmt == Fmt::BC4 || j.fmt == Fmt::BC5;
        const int64_t tight = (int64_t)(keep ? (source->width + 3) / 4 : source->width / 4) * ((j.fmt == Fmt::BC1 || j.fmt == Fmt::BC4) ? 8 : 16);
        if (block_row_pitch != tight) itw::fail_msg("itwCompressImageSlicedEx: block_row_pitch %lld != %lld (tight)", (long long)block_row_pitch, (long long)tight);
        done = compress_sliced(j, source, target, slices_of(source, slice_pixels), progress, user);
    });
    return ok && done;
}

void itwSetSliceWindow(int slices) { g_slice_window.store(slices < 0 ? -1 : slices, std::memory_order_relaxed); }

int itwSliceWindow(int dxgi_format, int width, int height, int64_t slice_pixels) { return itwSliceWindowFor(dxgi_format, nullptr, width, height, slice_pixels); }

int itwSliceWindowFor(int dxgi_format, const void* settings, int width, int height, int64_t slice_pixels)
{
    if (slice_window_setting() < 0) return 0;
    if (slice_pixels <= 0) slice_pixels = 0x40000;
    int64_t slices = ((int64_t)width * height) / slice_pixels;
    if (slices < 1) slices = 1;
    if (slices > (1 << 24)) slices = 1 << 24;
    const bool keep = dxgi_format == ITW_DXGI_FORMAT_BC4_UNORM || dxgi_format == ITW_DXGI_FORMAT_BC5_UNORM;
    const int64_t blocks = keep ? (int64_t)((width + 3) / 4) * ((height + 3) / 4) : (int64_t)(width / 4) * (height / 4);
    const bool heavy = dxgi_format == ITW_DXGI_FORMAT_BC7_UNORM || dxgi_format == ITW_DXGI_FORMAT_BC7_UNORM_SRGB ||
                       dxgi_format == ITW_DXGI_FORMAT_BC6H_UF16 || dxgi_format == ITW_DXGI_FORMAT_BC6H_SF16;
    const bool bc7 = dxgi_format == ITW_DXGI_FORMAT_BC7_UNORM || dxgi_format == ITW_DXGI_FORMAT_BC7_UNORM_SRGB;
    const bool every_shape = bc7 && settings && itw::bc7_scans_every_shape(*static_cast<const bc7_enc_settings*>(settings));
    return slice_window(heavy ? Fmt::BC7 : Fmt::BC1, blocks / slices, (int)slices, every_shape);
}

void  itwSetStream(void* s) { tls.user_stream = (hipStream_t)s; }
void* itwGetStream(void)    { return (void*)tls.user_stream; }

int itwAvailable(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void)hipGetLastError(); return 0; }
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return std::strncmp(p.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;     // the only code object this library carries
}

void itwSetErrorMode(int mode) { itw::g_error_mode.store(mode == ITW_ON_ERROR_RETURN ? 1 : 0); }
const char* itwLastError(void) { return itw::t_has_error ? itw::t_last_error : nullptr; }
void itwClearError(void) { itw::clear_failure(); }

const char* itwDeviceInfo(void)
{
    tls.info[0] = 0;
    itw::guarded([&] {
        int dev = 0;
        ITW_CHECK(hipGetDevice(&dev));
        hipDeviceProp_t p;
        ITW_CHECK(hipGetDeviceProperties(&p, dev));
        std::snprintf(tls.info, sizeof(tls.info), "%s / %s / %d CUs / %.0f MHz", p.gcnArchName, p.name,
                      p.multiProcessorCount, p.clockRate / 1000.0);
    });
    return tls.info;
}

void itwSetBc7Path(int path) { itw::set_bc7_path(path); itw::set_bc6h_path(path); }
void itwSetBc7Pilot(int percent) { itw::set_bc7_pilot(percent); }

const char* itwVersion(void) { return "itw-amd 0.1 gfx950 arith=x86-lut-nr contract=off"; }

int64_t itwBandForPart(int32_t width, int32_t height, int32_t bytes_per_block,
                       int32_t part, int32_t parts, int32_t* first_row, int32_t* row_count)
{
    const int64_t R = height / 4, bx = width / 4;
    if (parts <= 0 || part < 0 || part >= parts) { if (first_row) *first_row = 0; if (row_count) *row_count = 0; return -1; }
    const int64_t r0 = R * part / parts, r1 = R * (part + 1) / parts;
    if (first_row) *first_row = (int32_t)(r0 * 4);
    if (row_count) *row_count = (int32_t)((r1 - r0) * 4);
    return r0 * bx * bytes_per_block;
}

int64_t itwBandForPartEx(int32_t width, int32_t height, int32_t bytes_per_block, int32_t part, int32_t parts, int32_t keep_partial_blocks,
                         int32_t* first_row, int32_t* row_count)
{
    if (!keep_partial_blocks) return itwBandForPart(width, height, bytes_per_block, part, parts, first_row, row_count);
    const int64_t R = ((int64_t)height + 3) / 4, bx = ((int64_t)width + 3) / 4;
    if (parts <= 0 || part < 0 || part >= parts || height < 1 || width < 1) { if (first_row) *first_row = 0; if (row_count) *row_count = 0; return -1; }
    const int64_t r0 = R * part / parts, r1 = R * (part + 1) / parts;
    const int64_t y0 = r0 * 4, y1 = (r1 * 4 < height) ? r1 * 4 : height;       // the band that holds the last block row keeps its partial rows
    if (first_row) *first_row = (int32_t)y0;
    if (row_count) *row_count = (int32_t)(y1 > y0 ? y1 - y0 : 0);
    return r0 * bx * bytes_per_block;
}

void itwWarmupBC45(void)
{
    itwClearError();
    itw::guarded([&] { bind_thread_to_current_device(); itw::warmup_bc45(); });
}

#ifdef ITW_TEST_HOOKS      // libispc_texcomp_test.so only (include/itw_test_hooks.h)
int itwTestBc45IndexTable(uint32_t* host_out)
{
    itwClearError();
    return itw::guarded([&] { if (!host_out) itw::fail_msg("null pointer"); itw::copy_bc45_index_table(host_out, tls.user_stream); }) ? 0 : -1;
}

void itwTestBc7TwoSubsetBounds(const rgba_surface* d_src, float* d_out)
{
    itwClearError();
    itw::guarded([&] {
        if (!d_src || !d_src->ptr || !d_out) itw::fail_msg("null pointer");
        itw::launch_bc7_test_bounds(d_src->ptr, d_src->stride, d_src->width, d_src->height, d_out, tls.user_stream);
        ITW_CHECK(hipGetLastError());
    });
}

void itwTestRcp(const float* in, float* out, int64_t n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_test_rcp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tls.user_stream, in, out, n);
    itw::guarded([&] { ITW_CHECK(hipGetLastError()); });
}
void itwTestRsqrt(const float* in, float* out, int64_t n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_test_rsqrt, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tls.user_stream, in, out, n);
    itw::guarded([&] { ITW_CHECK(hipGetLastError()); });
}
void itwTestF2I(const float* in, int32_t* out, int64_t n)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_test_f2i, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tls.user_stream, in, out, n);
    itw::guarded([&] { ITW_CHECK(hipGetLastError()); });
}
#endif // ITW_TEST_HOOKS

} // extern "C"
