// multigpu.hip -- itwCompressImageMultiGPU[Ex] (include/itw_multigpu.h): one surface over all GPUs of the node from ONE
// process, host code in C++: band per rank (win32Threads.cpp:217-231 on block rows), scatter of a device-resident
// surface by peer copies (or none: bands already resident on their devices), gather of the output bands to the owner of
// `output` by RCCL send/recv (or peer copies), a rank's band cut in two so that the upload of its second half and the gather
// of its first run under an encode.
//
// One persistent host thread per rank, bound to its device: CompressBlocks* keeps per-thread, per-device state (stream,
// BC7 workspace), so a rank's thread is the natural owner of its streams and staging buffers.
//
// Failure model.  A rank that fails must never leave another rank waiting in a collective:
//   * everything that can fail for lack of resources (streams, events, peer access, staging buffers) happens in a PREPARE
//     step; the ranks then meet at a host-side barrier and post transfers only if every rank is ready;
//   * a failure after that point (a launch, a copy) raises the call's abort flag and, on the RCCL transport, calls
//     ncclCommAbort on every communicator, which releases a peer blocked in ncclGroupEnd / hipStreamSynchronize; the
//     communicators are rebuilt by the next call.  Every use of a communicator handle happens under that communicator's
//     mutex with the abort flag re-checked inside, and the abort takes the same mutex (bounded wait: a rank that sits INSIDE
//     an RCCL call is exactly the one the abort has to release), so no thread enters RCCL with a handle that was freed;
//   * a failed rank drains its streams before it reports, so the next call never reuses buffers that still have work queued;
//   * any C++ exception ends the rank's work as a failure, and the submitting thread reports it through report_failure():
//     abort mode aborts loudly, return mode returns false with the message in itwLastError();
//   * a rank that neither fails nor proceeds is the watchdog's: the submitting thread, which only waits, aborts the call
//     when a rank has not posted its work within ITW_MULTIGPU_POST_TIMEOUT_S (the first RCCL send/recv of a process sets up
//     its connections inside ncclSend / ncclGroupEnd) or the call has not finished within ITW_MULTIGPU_TIMEOUT_S.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the symbols are resolved with dlsym on first use
#include <dlfcn.h>
#include <sched.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <future>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/itw_multigpu.h"
#include "../../include/itw_amd.h"
#ifdef ITW_TEST_HOOKS
#include "../../include/itw_test_hooks.h"
#endif
#include "host_rt.hpp"

namespace {

constexpr int MAX_RANKS = 64;
constexpr int MAX_PIECES = 8;          // pieces of a rank's share: the 2 halves of its band, or up to 8 interleaved sub-bands
using Clock = std::chrono::steady_clock;
double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    bool load()
    {
        if (lib) return true;
        for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) return false;
#define ITW_SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(lib, "nccl" #f)); if (!f) { dlclose(lib); lib = nullptr; return false; }
        ITW_SYM(CommInitAll) ITW_SYM(CommDestroy) ITW_SYM(CommAbort) ITW_SYM(CommCount) ITW_SYM(Send) ITW_SYM(Recv) ITW_SYM(GroupStart)
        ITW_SYM(GroupEnd) ITW_SYM(GetErrorString)
#undef ITW_SYM
        return true;
    }
};

#define ITW_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) itw::fail_msg("%s failed: %s", #expr, g.rccl.GetErrorString(r_)); } while (0)

struct Call {                      // one itwCompressImageMultiGPU call, shared by the rank threads
    rgba_surface input;
    uint8_t* output = nullptr;
    CompressionFunc* fn = nullptr;
    const rgba_surface* bands = nullptr;                   // resident bands (one per rank, on the rank's device) or nullptr
    int bpb = 16, texel_bytes = 4, ranks = 1;
    int interleave = 1;                                    // K: sub-bands per rank (1: one contiguous band per rank, cut in two halves)
    bool keep_partial = false;
    bool src_dev = false, dst_dev = false;
    int src_device = -1, dst_device = -1, dst_rank = -1;   // dst_rank: the rank (on dst_device) that posts the receives
    bool use_rccl = false;
    int fail_rank = -1, fail_stage = 0, stall_ms = 0;      // itwMultiGpuTestInjectFailure
};

// device-side timestamps of one rank and one call: [piece][begin, end] per activity
enum { T_UP = 0, T_ENC = 1, T_GATHER = 2, T_RECV = 3, T_KINDS = 4 };

struct RankCtx {
    int rank = 0, device = 0;
    std::thread th;
    hipStream_t enc = nullptr, xfer = nullptr;
    hipEvent_t ev[T_KINDS][MAX_PIECES][2] = {};            // [kind][piece][begin | end], timing enabled
    bool rec[T_KINDS][MAX_PIECES] = {};                    // pair recorded in the current call
    void* d_in = nullptr;  size_t in_cap = 0;
    void* d_out = nullptr; size_t out_cap = 0;
    bool peers_enabled = false;
    bool pending = false, failed = false;
    std::atomic<int> stage{0};                             // 0 idle / running, 1 prepared, 2 everything posted, 3 done
    std::atomic<double> posted_ms{0.0};             // written by the rank thread, read by the submitting thread (stats)
    itw_multigpu_rank_stats st{};
    char msg[384] = {0};
};

struct Group {
    std::mutex m, submit, abort_mu;
    std::condition_variable work, done, ready_cv;
    std::vector<RankCtx*> ranks;
    int devices = 1, outstanding = 0;
    int ready = 0;                      // ranks that finished PREPARE (ok or not) in the current call
    bool prepare_failed = false;        // some rank failed in PREPARE: nobody posts a transfer
    std::atomic<bool> abort{false};     // some rank failed after PREPARE (or the watchdog fired): the others stop posting work
    bool comms_aborted = false;         // ncclCommAbort was called: communicators are rebuilt by the next call (g.m)
    bool wedged = false;                // a rank thread never came back from a call: the group cannot be used again
    bool rccl_dead = false;             // ncclCommInitAll did not return: RCCL is not tried again in this process
    bool quit = false;
    Call call;
    Clock::time_point t0;
    Rccl rccl;
    int ncomms = 0;                                        // communicators currently alive (= the rank count they were built for)
    std::atomic<ncclComm_t> comms[MAX_RANKS];              // one per rank when RCCL is in use (ranks == distinct devices)
    std::timed_mutex comm_mu[MAX_RANKS];                   // guards every use of comms[i] against abort_transfers()
    const char* transport = "peer";
    char note[96] = {0};
    std::atomic<int> peer_links{0};     // directed device pairs with peer access enabled (xGMI instead of a bounce through the host)
    std::atomic<int> inject_rank{-1}, inject_stage{0}, inject_stall{0};
    std::atomic<int> interleave{-1};    // requested sub-bands per rank (itwMultiGpuSetInterleave); -1: ITW_MULTIGPU_INTERLEAVE or the default, 4
    Group() { for (auto& c : comms) c.store(nullptr); }
};
Group& g = *new Group;             // never destroyed: rank threads outlive static destruction

int device_count()
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    return n;
}

void* grow(void*& buf, size_t& cap, size_t need)
{
    if (need > cap) {
        if (buf) { void* old = buf; buf = nullptr; cap = 0; ITW_CHECK(hipFree(old)); }
        void* fresh = nullptr;
        ITW_CHECK(hipMalloc(&fresh, need + need / 8 + 4096));
        buf = fresh; cap = need + need / 8 + 4096;
    }
    return buf;
}

int device_of(const void* p)
{
    hipPointerAttribute_t a;
    std::memset(&a, 0, sizeof a);
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) ? a.device : -1;
}

// block rows [r0, r1) of rank r (itwBandForPart's rule), then the half-band cut
void band_rows(int by, int rank, int ranks, int& r0, int& r1) { r0 = (int)((int64_t)by * rank / ranks); r1 = (int)((int64_t)by * (rank + 1) / ranks); }
void half_cut(int r0, int r1, int (&cut)[3]) { cut[0] = r0; cut[1] = (r1 - r0 >= 2) ? r0 + (r1 - r0 + 1) / 2 : r1; cut[2] = r1; }

// The pieces rank `rank` encodes, in posting order: block rows [cut[s][0], cut[s][1]).
//   K = 1  the reference's partition (win32Threads.cpp:217-231 on block rows): one contiguous band per rank, cut in two halves so that
//          the upload of the second and the gather of the first run under an encode;
//   K > 1  (round 5) the surface is cut into K * ranks sub-bands (itwBandForPart(j, K * ranks)) and sub-band j belongs to rank j % ranks:
//          since the bounded BC7 order a band's encode time depends on its content (1.4x between noise and a photograph), and K
//          interleaved sub-bands give every rank a sample of the whole surface.  The output is where it always was, so a rank's gather
//          is K contiguous runs instead of one.
int pieces_of(int by, int rank, int ranks, int K, int (&cut)[MAX_PIECES][2])
{
    if (K <= 1) {
        int r0, r1, h[3];
        band_rows(by, rank, ranks, r0, r1);
        half_cut(r0, r1, h);
        cut[0][0] = h[0]; cut[0][1] = h[1]; cut[1][0] = h[1]; cut[1][1] = h[2];
        return 2;
    }
    for (int s = 0; s < K; s++) band_rows(by, s * ranks + rank, K * ranks, cut[s][0], cut[s][1]);
    return K;
}
// K as a call uses it: the requested value where every sub-band still has at least 16 block rows, else 1
int effective_interleave(int by, int ranks, int K)
{
    if (K > MAX_PIECES) K = MAX_PIECES;
    return (K > 1 && (int64_t)by >= (int64_t)16 * K * ranks) ? K : 1;
}

int requested_interleave()
{
    int k = g.interleave.load();
    if (k < 0) {
        const char* e = std::getenv("ITW_MULTIGPU_INTERLEAVE");
        k = e ? std::atoi(e) : 4;
        k = k < 1 ? 1 : (k > MAX_PIECES ? MAX_PIECES : k);
        g.interleave.store(k);
    }
    return k;
}

// Peer access from this rank's device to every other device, once per rank thread: without it hipMemcpyPeerAsync and
// hipMemcpy2DAsync(hipMemcpyDefault) between two GPUs may stage through host memory instead of using xGMI.
void enable_peers(RankCtx& c)
{
    if (c.peers_enabled) return;
    c.peers_enabled = true;
    for (int d = 0; d < g.devices; d++) {
        if (d == c.device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, c.device, d) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (!can) continue;
        const hipError_t e = hipDeviceEnablePeerAccess(d, 0);
        if (e == hipSuccess) g.peer_links.fetch_add(1);
        else (void)hipGetLastError();                          // already enabled by another rank on this device / by the host: fine
    }
}

// Releases every rank that waits for a peer in this invocation: in the PREPARE barrier, or inside an RCCL call.  Called by the
// first rank that fails after PREPARE and by the watchdog.  ncclCommAbort frees the communicator, so it runs under the
// communicator's mutex (every enqueue holds it and re-checks the abort flag inside); if the mutex cannot be had for two seconds
// its holder sits inside RCCL -- the very thread the abort must release -- and the abort goes ahead.
void abort_transfers(const Call& k)
{
    g.abort.store(true);
    { std::lock_guard<std::mutex> lk(g.m); g.ready_cv.notify_all(); }
    if (!k.use_rccl) return;
    std::lock_guard<std::mutex> one(g.abort_mu);
    for (int i = 0; i < MAX_RANKS; i++) {
        if (!g.comms[i].load()) continue;
        const bool locked = g.comm_mu[i].try_lock_for(std::chrono::seconds(2));
        ncclComm_t c = g.comms[i].exchange(nullptr);
        if (c) (void)g.rccl.CommAbort(c);
        if (locked) g.comm_mu[i].unlock();
    }
    std::lock_guard<std::mutex> lk(g.m);
    g.comms_aborted = true;
}

// One enqueue on rank `rank`'s communicator: handle read and used under its mutex, abort flag re-checked inside.
template <class F>
void with_comm(int rank, F&& f)
{
    std::lock_guard<std::timed_mutex> lk(g.comm_mu[rank]);
    if (g.abort.load()) itw::fail_msg("stopped: another rank failed");
    ncclComm_t c = g.comms[rank].load();
    if (!c) itw::fail_msg("stopped: the communicator was aborted");
    f(c);
}

void drain(RankCtx& c) noexcept
{
    if (c.enc) (void)hipStreamSynchronize(c.enc);
    if (c.xfer) (void)hipStreamSynchronize(c.xfer);
    (void)hipGetLastError();
}

void mark(RankCtx& c, int kind, int piece, int end, hipStream_t st)
{
    ITW_CHECK(hipEventRecord(c.ev[kind][piece][end], st));
    if (end) c.rec[kind][piece] = true;
}

// rows [y0, y1) of the input surface -> pitched staging on this GPU.  A signed / overlapping stride (bottom-up surfaces: the
// reference indexes ptr + y*stride with a signed stride, kernel.ispc:105-151) cannot be a pitched copy: row by row, as
// CompressBlocks* does for one GPU (abi.hip).
void upload_rows(uint8_t* dpos, size_t pitch, const Call& k, int64_t y0, int64_t y1, size_t row_bytes, hipStream_t st)
{
    const uint8_t* src = k.input.ptr + y0 * (int64_t)k.input.stride;
    const hipMemcpyKind kind = k.src_dev ? hipMemcpyDefault : hipMemcpyHostToDevice;   // Default: peer copy from the owner GPU over xGMI
    if ((int64_t)k.input.stride >= (int64_t)row_bytes) {
        ITW_CHECK(hipMemcpy2DAsync(dpos, pitch, src, (size_t)k.input.stride, row_bytes, (size_t)(y1 - y0), kind, st));
    } else {
        for (int64_t y = 0; y < y1 - y0; y++)
            ITW_CHECK(hipMemcpyAsync(dpos + (size_t)y * pitch, src + y * (int64_t)k.input.stride, row_bytes, kind, st));
    }
}

// the rank's account of the call, from whatever event pairs were recorded (all complete: the streams were synchronised or drained)
void account(RankCtx& c) noexcept
{
    float sums[T_KINDS] = {0, 0, 0, 0}, last = 0.f;
    hipEvent_t first = nullptr;
    for (int kind = 0; kind < T_KINDS && !first; kind++) for (int h = 0; h < MAX_PIECES && !first; h++) if (c.rec[kind][h]) first = c.ev[kind][h][0];
    for (int kind = 0; kind < T_KINDS; kind++)
        for (int h = 0; h < MAX_PIECES; h++) {
            if (!c.rec[kind][h]) continue;
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, c.ev[kind][h][0], c.ev[kind][h][1]) == hipSuccess) sums[kind] += ms;
            if (first && hipEventElapsedTime(&ms, first, c.ev[kind][h][1]) == hipSuccess && ms > last) last = ms;
            // (the very first event recorded is the band's first upload or, without uploads, its first encode: nothing starts earlier)
        }
    (void)hipGetLastError();
    c.st.upload_ms = sums[T_UP]; c.st.encode_ms = sums[T_ENC]; c.st.gather_ms = sums[T_GATHER] + sums[T_RECV]; c.st.span_ms = last;
}

void run_rank(RankCtx& c, const Call& k)
{
    const int w = k.input.width, h = k.input.height;
    const int bx = k.keep_partial ? (w + 3) / 4 : w / 4, by = k.keep_partial ? (h + 3) / 4 : h / 4;
    int cut[MAX_PIECES][2];
    const int np = pieces_of(by, c.rank, k.ranks, k.interleave, cut);
    int total_rows = 0;
    for (int s = 0; s < np; s++) total_rows += cut[s][1] > cut[s][0] ? cut[s][1] - cut[s][0] : 0;
    const bool idle = total_rows <= 0;
    const size_t row_bytes = (size_t)w * k.texel_bytes;
    const size_t pitch = (row_bytes + 15) & ~(size_t)15;
    const bool src_here = k.bands || (k.src_dev && k.src_device == c.device);       // texels already on this GPU: encoded in place
    const bool dst_here = k.dst_dev && k.dst_device == c.device;
    // texel rows [y0, y1) of piece s: the piece that holds the surface's last block row keeps a partial block row (BC4/BC5)
    auto rows_of = [&](int s, int64_t& y0, int64_t& y1) { y0 = (int64_t)cut[s][0] * 4; y1 = (cut[s][1] == by) ? h : (int64_t)cut[s][1] * 4; };
    int64_t texel_rows = 0;
    for (int s = 0; s < np; s++) { int64_t y0, y1; rows_of(s, y0, y1); if (cut[s][1] > cut[s][0]) texel_rows += y1 - y0; }
    uint8_t* in = nullptr;
    uint8_t* out = nullptr;
    std::memset(c.rec, 0, sizeof c.rec);
    c.st = itw_multigpu_rank_stats{};
    c.st.rank = c.rank; c.st.device = c.device; c.st.block_row0 = cut[0][0]; c.st.block_rows = total_rows;

    // ---- PREPARE: everything that can fail for lack of resources, before any transfer is posted ----
    itw::Failure early;
    bool early_failed = false;
    try {
        if (!c.enc) {
            ITW_CHECK(hipStreamCreateWithFlags(&c.enc, hipStreamNonBlocking));
            ITW_CHECK(hipStreamCreateWithFlags(&c.xfer, hipStreamNonBlocking));
            for (auto& kind : c.ev) for (auto& piece : kind) for (auto& e : piece) ITW_CHECK(hipEventCreate(&e));
        }
        enable_peers(c);
        if (!idle) {
            if (k.bands) {
                // resident input: K = 1: surface `rank` holds the rank's band; K > 1: surface s * ranks + rank holds its s-th sub-band
                for (int s = 0; s < (k.interleave > 1 ? np : 1); s++) {
                    const int idx = k.interleave > 1 ? s * k.ranks + c.rank : c.rank;
                    int64_t y0, y1;
                    if (k.interleave > 1) { if (cut[s][1] <= cut[s][0]) continue; rows_of(s, y0, y1); }
                    else { y0 = (int64_t)cut[0][0] * 4; y1 = (cut[np - 1][1] == by) ? h : (int64_t)cut[np - 1][1] * 4; }
                    const rgba_surface& b = k.bands[idx];
                    if (!b.ptr || b.width != w || (int64_t)b.height < y1 - y0)
                        itw::fail_msg("resident band %d: %dx%d texels at %p, expected %d x %lld", idx, b.width, b.height, (void*)b.ptr, w, (long long)(y1 - y0));
                    if (device_of(b.ptr) != c.device) itw::fail_msg("resident band %d is not on device %d (the device rank %d runs on)", idx, c.device, c.rank);
                }
            }
            in = src_here ? nullptr : (uint8_t*)grow(c.d_in, c.in_cap, pitch * (size_t)texel_rows);
            out = dst_here ? nullptr : (uint8_t*)grow(c.d_out, c.out_cap, (size_t)total_rows * bx * k.bpb);
        }
        if (k.fail_rank == c.rank && k.fail_stage == 1) itw::fail_msg("injected failure in PREPARE (itwMultiGpuTestInjectFailure)");
    } catch (const itw::Failure& f) { early = f; early_failed = true; }
    catch (...) { std::snprintf(early.msg, sizeof early.msg, "unexpected C++ exception in PREPARE"); early_failed = true; }   // the barrier below must be reached
    {
        std::unique_lock<std::mutex> lk(g.m);
        if (early_failed) g.prepare_failed = true;
        if (++g.ready == k.ranks) g.ready_cv.notify_all();
        else g.ready_cv.wait(lk, [&] { return g.ready >= k.ranks || g.abort.load(); });
        if (early_failed) throw early;
        if (g.prepare_failed) return;                    // another rank cannot take part: nobody sends, nobody waits
        if (g.ready < k.ranks) itw::fail_msg("stopped: another rank never finished preparing (watchdog)");
    }
    c.stage.store(1);
    if (k.fail_rank == c.rank && k.fail_stage == 3)      // a rank that neither fails nor proceeds: the watchdog's case
        for (int waited = 0; waited < k.stall_ms && !g.abort.load(); waited += 20) std::this_thread::sleep_for(std::chrono::milliseconds(20));
    if (idle) { c.posted_ms.store(ms_since(g.t0)); c.stage.store(2); return; }

    // ---- TRANSFER + ENCODE ----
    // Posting order (host side): upload 0, encode 0 | upload s on the transfer stream, encode s | ... | then every gather on the transfer stream,
    // gather s behind encode s's event.  What overlaps ON THE DEVICE:
    //   * the encodes are launches (asynchronous: posting all of them takes well under a millisecond against tens of milliseconds of work), so
    //     with device-side destinations every gather is queued long before its encode ends and gather s runs under encode s+1;
    //   * the transfer stream is ONE stream: gather 0 sits behind upload K-1 in it.  Uploads from pageable host memory are done when their call
    //     returns (they block the posting thread, which is why encode s is not launched before upload s is in), peer-copy uploads from another
    //     GPU finish under encode 0..s-1 -- either way no gather waits for an upload that is still needed by a later encode;
    //   * a download into pageable host memory blocks the posting thread until the copy is done: posted between the encodes it would hold back
    //     the next launch, hence "gathers last".  With host memory on both sides the K pieces cost K blocking uploads + K blocking downloads;
    //     each still overlaps its neighbour's encode on the device.
    try {
        itwSetStream(c.enc);
        uint8_t* o[MAX_PIECES] = {nullptr};
        size_t nbytes[MAX_PIECES] = {0};
        size_t in_rows_done = 0, out_rows_done = 0;
        for (int s = 0; s < np; s++) {
            const int a = cut[s][0], b = cut[s][1];
            if (b <= a) continue;
            if (g.abort.load()) itw::fail_msg("stopped: another rank failed");
            int64_t y0, y1;
            rows_of(s, y0, y1);
            rgba_surface sub = k.input;
            sub.height = (int)(y1 - y0);
            if (k.bands) {
                const rgba_surface& band = k.bands[k.interleave > 1 ? s * k.ranks + c.rank : c.rank];
                const int64_t band_first = k.interleave > 1 ? y0 : (int64_t)cut[0][0] * 4;
                sub.ptr = band.ptr + (y0 - band_first) * (int64_t)band.stride;
                sub.stride = band.stride;
            } else if (src_here) {
                sub.ptr = k.input.ptr + y0 * (int64_t)k.input.stride;
            } else {
                // host -> this GPU over its own PCIe link, or owner GPU -> this GPU over xGMI; later pieces on the transfer stream
                uint8_t* dpos = in + in_rows_done * pitch;
                const hipStream_t up = (s == 0) ? c.enc : c.xfer;
                mark(c, T_UP, s, 0, up);
                upload_rows(dpos, pitch, k, y0, y1, row_bytes, up);
                mark(c, T_UP, s, 1, up);
                if (s > 0) ITW_CHECK(hipStreamWaitEvent(c.enc, c.ev[T_UP][s][1], 0));
                sub.ptr = dpos;
                sub.stride = (int32_t)pitch;
                in_rows_done += (size_t)(y1 - y0);
            }
            o[s] = dst_here ? k.output + (size_t)a * bx * k.bpb : out + out_rows_done * bx * k.bpb;     // (resident output: encoded in place)
            nbytes[s] = (size_t)(b - a) * bx * k.bpb;
            out_rows_done += (size_t)(b - a);
            itwClearError();
            mark(c, T_ENC, s, 0, c.enc);
            k.fn(&sub, o[s]);                                                         // device pointers: asynchronous on c.enc
            if (const char* e = itwLastError()) itw::fail_msg("%s", e);
            mark(c, T_ENC, s, 1, c.enc);
        }
        if (!dst_here) {                                                              // (else: encoded in place)
            bool first_posted = false;
            for (int s = 0; s < np; s++) {
                if (!nbytes[s]) continue;
                if (g.abort.load()) itw::fail_msg("stopped: another rank failed");
                ITW_CHECK(hipStreamWaitEvent(c.xfer, c.ev[T_ENC][s][1], 0));
                uint8_t* dpos = k.output + (size_t)cut[s][0] * bx * k.bpb;
                mark(c, T_GATHER, s, 0, c.xfer);
                if (!k.dst_dev)      ITW_CHECK(hipMemcpyAsync(dpos, o[s], nbytes[s], hipMemcpyDeviceToHost, c.xfer));
                else if (k.use_rccl) with_comm(c.rank, [&](ncclComm_t comm) { ITW_NCCL(g.rccl.Send(o[s], nbytes[s], ncclUint8, k.dst_rank, comm, c.xfer)); });
                else                 ITW_CHECK(hipMemcpyPeerAsync(dpos, k.dst_device, o[s], c.device, nbytes[s], c.xfer));
                mark(c, T_GATHER, s, 1, c.xfer);
                if (!first_posted && k.fail_rank == c.rank && k.fail_stage == 2) itw::fail_msg("injected failure after the first piece (itwMultiGpuTestInjectFailure)");
                first_posted = true;
            }
        } else if (k.fail_rank == c.rank && k.fail_stage == 2) {
            itw::fail_msg("injected failure after the first piece (itwMultiGpuTestInjectFailure)");
        }
        // the rank that owns `output` posts the matching receives, one group per piece index so pieces complete independently
        if (k.use_rccl && k.dst_dev && c.rank == k.dst_rank) {
            for (int s = 0; s < np; s++) {
                mark(c, T_RECV, s, 0, c.xfer);
                with_comm(c.rank, [&](ncclComm_t comm) {
                    // an abort that could not take this communicator's mutex within its two seconds frees the handle
                    // under us; looking at the flag and the handle again before every RCCL call keeps the exposure to the ONE call that
                    // is in flight at that moment (which is the call the abort exists to release)
                    auto still_ours = [&] { if (g.abort.load() || g.comms[c.rank].load() != comm) itw::fail_msg("stopped: another rank failed"); };
                    still_ours();
                    ITW_NCCL(g.rccl.GroupStart());
                    for (int p = 0; p < k.ranks; p++) {
                        still_ours();
                        if (p == c.rank) continue;
                        int pc[MAX_PIECES][2];
                        const int pn = pieces_of(by, p, k.ranks, k.interleave, pc);
                        if (s >= pn || pc[s][1] <= pc[s][0]) continue;
                        ITW_NCCL(g.rccl.Recv(k.output + (size_t)pc[s][0] * bx * k.bpb, (size_t)(pc[s][1] - pc[s][0]) * bx * k.bpb, ncclUint8, p, comm, c.xfer));
                    }
                    still_ours();
                    ITW_NCCL(g.rccl.GroupEnd());
                });
                mark(c, T_RECV, s, 1, c.xfer);
            }
        }
        c.posted_ms.store(ms_since(g.t0));
        c.stage.store(2);
        ITW_CHECK(hipStreamSynchronize(c.enc));
        ITW_CHECK(hipStreamSynchronize(c.xfer));
        if (g.abort.load()) itw::fail_msg("stopped: another rank failed");           // an aborted communicator completes its streams without data
    } catch (const itw::Failure&) {
        abort_transfers(k);                          // nobody may keep waiting for this rank's sends / receives
        drain(c);                                    // nothing of this call stays queued on buffers the next call reuses
        throw;
    }
}

// Best effort: run the rank's host thread on the CPUs of its GPU's NUMA node, so that the pageable band uploads / downloads of the
// host-pointer case stay on the socket the GPU hangs off.  PCI bus id -> sysfs numa_node -> that node's
// cpulist, INTERSECTED with the affinity the thread inherited (taskset / numactl / the job scheduler stay in charge: ADVICE r03)
// -> sched_setaffinity.  Anything missing (no sysfs, node -1, one-node box, empty intersection): the thread keeps its affinity.
// ITW_MULTIGPU_AFFINITY=0 disables it.
void place_thread_near_device(int device)
{
    const char* e = std::getenv("ITW_MULTIGPU_AFFINITY");
    if (e && e[0] == '0') return;
    char bus[32] = {0};
    if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) { (void)hipGetLastError(); return; }
    for (char* p = bus; *p; p++) if (*p >= 'A' && *p <= 'Z') *p = (char)(*p - 'A' + 'a');          // sysfs spells the id in lower case
    char path[160];
    std::snprintf(path, sizeof path, "/sys/bus/pci/devices/%s/numa_node", bus);
    int node = -1;
    if (FILE* f = std::fopen(path, "r")) { if (std::fscanf(f, "%d", &node) != 1) node = -1; std::fclose(f); }
    if (node < 0) return;
    std::snprintf(path, sizeof path, "/sys/devices/system/node/node%d/cpulist", node);
    char list[1024] = {0};
    if (FILE* f = std::fopen(path, "r")) { if (!std::fgets(list, (int)sizeof list, f)) list[0] = 0; std::fclose(f); }
    cpu_set_t inherited, set;
    CPU_ZERO(&inherited);
    if (sched_getaffinity(0, sizeof inherited, &inherited) != 0) return;
    CPU_ZERO(&set);
    int any = 0;
    for (const char* p = list; *p;) {                            // "0-31,64-95"
        char* q = nullptr;
        const long a = std::strtol(p, &q, 10);
        if (q == p) break;
        long b = a;
        if (*q == '-') { p = q + 1; b = std::strtol(p, &q, 10); }
        for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; cpu++)
            if (CPU_ISSET((int)cpu, &inherited)) { CPU_SET((int)cpu, &set); any = 1; }
        p = (*q == ',') ? q + 1 : q;
        if (*q != ',' ) break;
    }
    if (any) (void)sched_setaffinity(0, sizeof set, &set);
}

void rank_main(RankCtx* c)
{
    (void)hipSetDevice(c->device);
    place_thread_near_device(c->device);
    std::unique_lock<std::mutex> lk(g.m);
    for (;;) {
        g.work.wait(lk, [&] { return g.quit || c->pending; });
        if (g.quit) return;
        const Call k = g.call;
        lk.unlock();
        itw::Failure fail;
        bool bad = false;
        try { run_rank(*c, k); }
        catch (const itw::Failure& f) { fail = f; bad = true; }
        catch (const std::exception& e) { std::snprintf(fail.msg, sizeof fail.msg, "C++ exception: %s", e.what()); bad = true; abort_transfers(k); drain(*c); }
        catch (...) { std::snprintf(fail.msg, sizeof fail.msg, "unexpected C++ exception"); bad = true; abort_transfers(k); drain(*c); }
        account(*c);
        c->stage.store(3);
        lk.lock();
        c->pending = false;
        c->failed = bad;
        if (bad) std::snprintf(c->msg, sizeof c->msg, "rank %d (device %d): %.340s", c->rank, c->device, fail.msg);
        if (--g.outstanding == 0) g.done.notify_all();
    }
}

void ensure_ranks(int n)          // g.submit held
{
    std::lock_guard<std::mutex> lk(g.m);
    g.devices = device_count();
    while ((int)g.ranks.size() < n) {
        RankCtx* c = new RankCtx;
        c->rank = (int)g.ranks.size();
        c->device = c->rank % g.devices;
        g.ranks.push_back(c);
        c->th = std::thread(rank_main, c);
    }
}

int env_seconds(const char* name, int dflt)
{
    const char* e = std::getenv(name);
    const int v = e ? std::atoi(e) : 0;
    return v > 0 ? v : dflt;
}

// RCCL is usable for a call when every rank sits on its own device (rank r on device r); communicators are (re)built when the
// rank count changes or after a call that had to abort them.  Says why not in g.note.  ncclCommInitAll runs on a helper thread so
// that a hang inside it (fabric manager, IPC) ends as "peer transport" instead of a hung host application.
bool prepare_rccl(int ranks)
{
    auto no = [](const char* why) { std::snprintf(g.note, sizeof g.note, "%s", why); return false; };
    const char* t = std::getenv("ITW_MULTIGPU_TRANSPORT");
    if (t && !std::strcmp(t, "peer")) return no("ITW_MULTIGPU_TRANSPORT=peer");
    if (ranks > g.devices) return no("more ranks than devices: ranks share a device, peer copies");
    if (g.rccl_dead) return no("ncclCommInitAll did not return earlier in this process");
    if (!g.rccl.load()) return no("librccl.so could not be loaded");
    if (g.comms_aborted) { g.ncomms = 0; g.comms_aborted = false; }        // ncclCommAbort released them already (handles are null)
    if (g.ncomms == ranks) return true;
    for (int i = 0; i < MAX_RANKS; i++) if (ncclComm_t c = g.comms[i].exchange(nullptr)) (void)g.rccl.CommDestroy(c);
    g.ncomms = 0;
    struct Init { std::vector<ncclComm_t> comms; std::vector<int> devs; ncclResult_t rc = ncclSuccess; };
    auto init = std::make_shared<Init>();
    init->comms.assign((size_t)ranks, nullptr);
    init->devs.resize((size_t)ranks);
    for (int i = 0; i < ranks; i++) init->devs[(size_t)i] = i;
    auto fin = std::make_shared<std::promise<void>>();
    std::future<void> fut = fin->get_future();
    const auto fn = g.rccl.CommInitAll;
    std::thread([init, fin, fn, ranks] { init->rc = fn(init->comms.data(), ranks, init->devs.data()); fin->set_value(); }).detach();
    if (fut.wait_for(std::chrono::seconds(env_seconds("ITW_MULTIGPU_INIT_TIMEOUT_S", 120))) != std::future_status::ready) {
        g.rccl_dead = true;
        return no("ncclCommInitAll did not return in time");
    }
    if (init->rc != ncclSuccess) { std::snprintf(g.note, sizeof g.note, "ncclCommInitAll failed: %.60s", g.rccl.GetErrorString(init->rc)); return false; }
    for (int i = 0; i < ranks; i++) g.comms[i].store(init->comms[(size_t)i]);
    g.ncomms = ranks;
    return true;
}

// n_bands: how many surfaces `bands` holds -- K * ranks for K sub-bands per rank, stated by the caller per call (the library cannot see
// the array's length); 0 = the round-4 contract: `ranks` surfaces, band r on rank r (K = 1), whatever the process-wide interleave is.
bool compress_multi(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks,
                    const rgba_surface* bands, int n_bands, itw_multigpu_stats* stats)
{
    bool ok = false;
    itwClearError();
    if (stats) std::memset(stats, 0, sizeof *stats);
    itw::guarded([&] {
        if (!input || (!input->ptr && !bands) || !output || !cmpFunc) itw::fail_msg("itwCompressImageMultiGPU: null argument");
        if (bands && ranks <= 0) itw::fail_msg("itwCompressImageMultiGPUEx: resident bands need an explicit rank count");
        std::lock_guard<std::mutex> one(g.submit);
        if (g.wedged) itw::fail_msg("itwCompressImageMultiGPU: a rank thread never returned from an earlier call (watchdog); restart the process");
        Call k;
        k.input = *input; k.output = output; k.fn = cmpFunc; k.bands = bands;
        k.keep_partial = dxgi_format == ITW_DXGI_FORMAT_BC4_UNORM || dxgi_format == ITW_DXGI_FORMAT_BC5_UNORM;
        k.bpb = GetBytesPerBlock(dxgi_format);
        k.texel_bytes = (dxgi_format == ITW_DXGI_FORMAT_BC6H_UF16 || dxgi_format == ITW_DXGI_FORMAT_BC6H_SF16) ? 8 : 4;
        const int by = k.keep_partial ? (input->height + 3) / 4 : input->height / 4;
        if (by <= 0 || input->width < (k.keep_partial ? 1 : 4)) { ok = true; return; }
        int n = ranks > 0 ? ranks : itwMultiGpuRanks();
        n = n > MAX_RANKS ? MAX_RANKS : n;
        if (bands && n > by) itw::fail_msg("itwCompressImageMultiGPUEx: %d resident bands for %d block rows", n, by);
        n = n > by ? by : n;
        k.ranks = n;
        if (bands) {
            // resident input: K comes with the call, never from process-wide state another thread may change between the caller's
            // itwMultiGpuPieces() and this call
            k.interleave = 1;
            if (n_bands > 0) {
                if (n != ranks) itw::fail_msg("itwCompressImageMultiGPUBands: %d ranks asked for, %d usable (at most %d, and one block row each)", ranks, n, MAX_RANKS);
                if (n_bands % n != 0 || n_bands / n > MAX_PIECES) itw::fail_msg("itwCompressImageMultiGPUBands: %d resident sub-bands for %d ranks (K = 1..%d per rank)", n_bands, n, MAX_PIECES);
                if (n_bands > by) itw::fail_msg("itwCompressImageMultiGPUBands: %d resident sub-bands for %d block rows", n_bands, by);
                k.interleave = n_bands / n;
            }
        } else k.interleave = effective_interleave(by, n, requested_interleave());
        ensure_ranks(n);
        k.src_device = bands ? -1 : device_of(input->ptr); k.src_dev = k.src_device >= 0;
        k.dst_device = device_of(output);     k.dst_dev = k.dst_device >= 0;
        k.dst_rank = k.dst_dev ? k.dst_device % g.devices : -1;                  // rank r lives on device r % devices: the lowest one there
        if (k.dst_dev && k.dst_rank >= n) k.dst_rank = -1;                        // no rank on the owner: peer copies only
        g.note[0] = 0;
        if (!k.dst_dev) std::snprintf(g.note, sizeof g.note, "output in host memory: every GPU downloads its own band");
        else if (n == 1) std::snprintf(g.note, sizeof g.note, "one rank: nothing to gather");
        else if (k.dst_rank < 0) std::snprintf(g.note, sizeof g.note, "no rank runs on the device that owns the output: peer copies");
        k.use_rccl = k.dst_dev && k.dst_rank >= 0 && n > 1 && prepare_rccl(n);
        g.transport = k.use_rccl ? "rccl" : (k.dst_dev ? "peer" : "host");
        k.fail_rank = g.inject_rank.exchange(-1); k.fail_stage = g.inject_stage.exchange(0); k.stall_ms = g.inject_stall.exchange(0);
        {
            std::lock_guard<std::mutex> lk(g.m);
            g.call = k;
            g.t0 = Clock::now();
            g.ready = 0; g.prepare_failed = false; g.abort.store(false);
            for (int i = 0; i < n; i++) { RankCtx* c = g.ranks[(size_t)i]; c->pending = true; c->failed = false; c->stage.store(0); c->posted_ms.store(0.0); }
            g.outstanding = n;
        }
        g.work.notify_all();
        // The submitting thread only waits: it is the watchdog.
        const double post_limit = 1e3 * env_seconds("ITW_MULTIGPU_POST_TIMEOUT_S", 30), total_limit = 1e3 * env_seconds("ITW_MULTIGPU_TIMEOUT_S", 600);
        bool fired = false;
        double fired_at = 0.0;
        std::unique_lock<std::mutex> lk(g.m);
        while (g.outstanding != 0) {
            g.done.wait_for(lk, std::chrono::milliseconds(fired ? 100 : 250), [&] { return g.outstanding == 0; });
            if (g.outstanding == 0) break;
            const double t = ms_since(g.t0);
            if (!fired) {
                bool late = t > total_limit;
                if (t > post_limit) for (int i = 0; i < n; i++) late = late || g.ranks[(size_t)i]->stage.load() < 2;
                if (late) {
                    fired = true; fired_at = t;
                    lk.unlock();
                    abort_transfers(k);              // releases the PREPARE barrier and every rank blocked inside RCCL
                    lk.lock();
                }
            } else if (t > fired_at + 20e3) {
                g.wedged = true;                     // a thread stuck in the driver cannot be recovered from here
                break;
            }
        }
        const double wall = ms_since(g.t0);
        if (stats) {
            stats->ranks = n; stats->devices = g.devices; stats->peer_links = g.peer_links.load();
            stats->watchdog_fired = fired ? 1 : 0; stats->resident_bands = bands ? 1 : 0; stats->interleave = k.interleave;
            stats->wall_ms = (float)wall;
            std::snprintf(stats->transport, sizeof stats->transport, "%s", g.transport);
            std::snprintf(stats->transport_note, sizeof stats->transport_note, "%s", g.note);
            double posted = 0.0;
            for (int i = 0; i < n; i++) {
                const RankCtx* c = g.ranks[(size_t)i];
                if (c->stage.load() == 3) stats->rank[i] = c->st;
                else { stats->rank[i] = itw_multigpu_rank_stats{}; stats->rank[i].rank = i; stats->rank[i].device = c->device; }
                if (c->stage.load() >= 2 && c->posted_ms.load() > posted) posted = c->posted_ms.load();
            }
            stats->posted_ms = (float)posted;
            if (k.use_rccl) {
                ncclComm_t c0 = g.comms[k.dst_rank].load();
                int cnt = 0;
                if (c0 && g.rccl.CommCount(c0, &cnt) == ncclSuccess) stats->rccl_ranks = cnt;
            }
        }
        if (g.wedged) {
            itw::Failure f;
            std::snprintf(f.msg, sizeof f.msg, "itwCompressImageMultiGPU: watchdog: a rank did not come back %.0f s after the call was aborted", (wall - fired_at) / 1e3);
            lk.unlock();
            throw f;
        }
        // report the rank that failed on its own, not one that merely stopped because of it
        int first = -1;
        for (int i = 0; i < n; i++)
            if (g.ranks[(size_t)i]->failed && (first < 0 || std::strstr(g.ranks[(size_t)first]->msg, "stopped: "))) first = i;
        if (first >= 0 || fired) {
            itw::Failure f;
            if (fired) std::snprintf(f.msg, sizeof f.msg, "watchdog: not every rank had posted its work after %.1f s (ITW_MULTIGPU_POST_TIMEOUT_S / _TIMEOUT_S); %.250s",
                                     fired_at / 1e3, first >= 0 ? g.ranks[(size_t)first]->msg : "");
            else std::snprintf(f.msg, sizeof f.msg, "%s", g.ranks[(size_t)first]->msg);
            lk.unlock();
            throw f;
        }
        ok = true;
    });
    return ok;
}

} // namespace

extern "C" {


void itwMultiGpuSetInterleave(int k) { g.interleave.store(k < 1 ? 1 : (k > MAX_PIECES ? MAX_PIECES : k)); }

int itwMultiGpuPieces(int32_t height, int ranks, int keep_partial_blocks)
{
    const int by = keep_partial_blocks ? (height + 3) / 4 : height / 4;
    if (ranks <= 0 || by <= 0) return 0;
    if (ranks > MAX_RANKS) ranks = MAX_RANKS;                  // exactly compress_multi's clamps
    return effective_interleave(by, ranks > by ? by : ranks, requested_interleave());
}

int itwMultiGpuRanks(void)
{
    const char* e = std::getenv("ITW_MULTIGPU_RANKS");
    const int n = e ? std::atoi(e) : 0;
    return n >= 1 ? (n > MAX_RANKS ? MAX_RANKS : n) : device_count();
}

const char* itwMultiGpuTransport(void) { return g.transport; }

int itwMultiGpuPeerLinks(void) { return g.peer_links.load(); }

#ifdef ITW_TEST_HOOKS      // libispc_texcomp_test.so only (include/itw_test_hooks.h); in the product nothing ever arms the injection
void itwMultiGpuTestInjectFailure(int rank, int stage, int stall_ms)
{
    g.inject_stage.store(stage); g.inject_stall.store(stall_ms); g.inject_rank.store(rank);
}
#endif

bool itwCompressImageMultiGPU(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks)
{
    return compress_multi(input, output, cmpFunc, dxgi_format, ranks, nullptr, 0, nullptr);
}

bool itwCompressImageMultiGPUEx(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks,
                                const rgba_surface* resident_bands, itw_multigpu_stats* stats)
{
    return compress_multi(input, output, cmpFunc, dxgi_format, ranks, resident_bands, 0, stats);
}

bool itwCompressImageMultiGPUBands(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks,
                                   const rgba_surface* resident_bands, int n_resident_bands, itw_multigpu_stats* stats)
{
    if (!resident_bands || n_resident_bands <= 0) {
        itwClearError();
        itw::guarded([&] { itw::fail_msg("itwCompressImageMultiGPUBands: no resident sub-bands (use itwCompressImageMultiGPUEx for a surface in one piece)"); });
        return false;
    }
    return compress_multi(input, output, cmpFunc, dxgi_format, ranks, resident_bands, n_resident_bands, stats);
}

} // extern "C"
