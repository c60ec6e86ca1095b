// multigpu.hip -- itwCompressImageMultiGPU (include/itw_multigpu.h): one surface over all GPUs of the node from ONE
// process, host code in C++: band per rank (win32Threads.cpp:217-231 on block rows), scatter of a device-resident
// surface by peer copies, gather of the output bands to the owner of `output` by RCCL send/recv (or peer copies), the
// gather of a rank's first half-band overlapping the encode of its second.
//
// One persistent host thread per rank, bound to its device: CompressBlocks* keeps per-thread, per-device state (stream,
// BC7 workspace), so a rank's thread is the natural owner of its streams and staging buffers.
//
// Failure model.  A rank that fails must never leave another rank waiting in a collective:
//   * everything that can fail for lack of resources (streams, events, peer access, staging buffers) happens in a PREPARE
//     step; the ranks then meet at a host-side barrier and post transfers only if every rank is ready;
//   * a failure after that point (a launch, a copy) raises the call's abort flag and, on the RCCL transport, calls
//     ncclCommAbort on every communicator, which releases a peer blocked in ncclGroupEnd / hipStreamSynchronize; the
//     communicators are rebuilt by the next call;
//   * a failed rank drains its streams before it reports, so the next call never reuses buffers that still have work queued;
//   * any C++ exception ends the rank's work as a failure, and the submitting thread reports it through report_failure():
//     abort mode aborts loudly, return mode returns false with the message in itwLastError().
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the symbols are resolved with dlsym on first use
#include <dlfcn.h>
#include <sched.h>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <mutex>
#include <thread>
#include <vector>
#include "../../include/itw_multigpu.h"
#include "../../include/itw_amd.h"
#include "host_rt.hpp"

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
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
        ITW_SYM(CommInitAll) ITW_SYM(CommDestroy) ITW_SYM(CommAbort) ITW_SYM(Send) ITW_SYM(Recv) ITW_SYM(GroupStart) ITW_SYM(GroupEnd) ITW_SYM(GetErrorString)
#undef ITW_SYM
        return true;
    }
};

#define ITW_NCCL(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) itw::fail_msg("%s failed: %s", #expr, g.rccl.GetErrorString(r_)); } while (0)

struct Call {                      // one itwCompressImageMultiGPU call, shared by the rank threads
    rgba_surface input;
    uint8_t* output = nullptr;
    CompressionFunc* fn = nullptr;
    int bpb = 16, texel_bytes = 4, ranks = 1;
    bool keep_partial = false;
    bool src_dev = false, dst_dev = false;
    int src_device = -1, dst_device = -1, dst_rank = -1;   // dst_rank: the rank (on dst_device) that posts the receives
    bool use_rccl = false;
    int fail_rank = -1, fail_stage = 0;                    // test hook (ITW_MULTIGPU_TEST_FAIL="rank:stage"): that rank throws in
};                                                         // stage 1 = prepare, 2 = after its first half-band was posted

struct RankCtx {
    int rank = 0, device = 0;
    std::thread th;
    hipStream_t enc = nullptr, xfer = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    void* d_in = nullptr;  size_t in_cap = 0;
    void* d_out = nullptr; size_t out_cap = 0;
    bool peers_enabled = false;
    bool pending = false, failed = false;
    char msg[384] = {0};
};

struct Group {
    std::mutex m, submit;
    std::condition_variable work, done, ready_cv;
    std::vector<RankCtx*> ranks;
    int devices = 1, outstanding = 0;
    int ready = 0;                      // ranks that finished PREPARE (ok or not) in the current call
    bool prepare_failed = false;        // some rank failed in PREPARE: nobody posts a transfer
    std::atomic<bool> abort{false};     // some rank failed after PREPARE: the others stop posting work
    bool comms_aborted = false;         // ncclCommAbort was called: communicators are rebuilt by the next call (g.m)
    bool quit = false;
    Call call;
    Rccl rccl;
    std::vector<ncclComm_t> comms;      // one per rank when RCCL is in use (ranks == distinct devices)
    const char* transport = "peer";
    std::atomic<int> peer_links{0};     // directed device pairs with peer access enabled (xGMI instead of a bounce through the host)
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
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return -1; }
    return (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) ? a.device : -1;
}

// block rows [r0, r1) of rank r (itwBandForPart's rule), then the half-band cut
void band_rows(int by, int rank, int ranks, int& r0, int& r1) { r0 = (int)((int64_t)by * rank / ranks); r1 = (int)((int64_t)by * (rank + 1) / ranks); }

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

// Releases every rank that waits in an RCCL call of this invocation.  Called by the first rank that fails after PREPARE.
void abort_transfers(const Call& k)
{
    g.abort.store(true);
    if (!k.use_rccl) return;
    std::lock_guard<std::mutex> lk(g.m);
    if (g.comms_aborted) return;
    g.comms_aborted = true;
    for (ncclComm_t c : g.comms) if (c) (void)g.rccl.CommAbort(c);
}

void drain(RankCtx& c) noexcept
{
    if (c.enc) (void)hipStreamSynchronize(c.enc);
    if (c.xfer) (void)hipStreamSynchronize(c.xfer);
    (void)hipGetLastError();
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

void run_rank(RankCtx& c, const Call& k)
{
    const int w = k.input.width, h = k.input.height;
    const int bx = k.keep_partial ? (w + 3) / 4 : w / 4, by = k.keep_partial ? (h + 3) / 4 : h / 4;
    int r0, r1;
    band_rows(by, c.rank, k.ranks, r0, r1);
    const bool idle = r1 <= r0;
    const size_t row_bytes = (size_t)w * k.texel_bytes;
    const size_t pitch = (row_bytes + 15) & ~(size_t)15;
    const bool src_here = k.src_dev && k.src_device == c.device;
    const bool dst_here = k.dst_dev && k.dst_device == c.device;
    const size_t band_out = idle ? 0 : (size_t)(r1 - r0) * bx * k.bpb;
    const int64_t first_row = (int64_t)r0 * 4;
    const int64_t last_row = (r1 == by) ? h : (int64_t)r1 * 4;                    // the last band keeps a partial block row (BC4/BC5)
    uint8_t* in = nullptr;
    uint8_t* out = nullptr;

    // ---- PREPARE: everything that can fail for lack of resources, before any transfer is posted ----
    itw::Failure early;
    bool early_failed = false;
    try {
        if (!c.enc) {
            ITW_CHECK(hipStreamCreateWithFlags(&c.enc, hipStreamNonBlocking));
            ITW_CHECK(hipStreamCreateWithFlags(&c.xfer, hipStreamNonBlocking));
            for (auto& e : c.ev) ITW_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        }
        enable_peers(c);
        if (!idle) {
            in = src_here ? nullptr : (uint8_t*)grow(c.d_in, c.in_cap, pitch * (size_t)(last_row - first_row));
            out = dst_here ? k.output + (size_t)r0 * bx * k.bpb : (uint8_t*)grow(c.d_out, c.out_cap, band_out);
        }
        if (k.fail_rank == c.rank && k.fail_stage == 1) itw::fail_msg("injected failure in PREPARE (ITW_MULTIGPU_TEST_FAIL)");
    } catch (const itw::Failure& f) { early = f; early_failed = true; }
    catch (...) { std::snprintf(early.msg, sizeof early.msg, "unexpected C++ exception in PREPARE"); early_failed = true; }   // the barrier below must be reached
    {
        std::unique_lock<std::mutex> lk(g.m);
        if (early_failed) g.prepare_failed = true;
        if (++g.ready == k.ranks) g.ready_cv.notify_all();
        else g.ready_cv.wait(lk, [&] { return g.ready >= k.ranks; });
        if (early_failed) throw early;
        if (g.prepare_failed) return;                    // another rank cannot take part: nobody sends, nobody waits
    }
    if (idle) return;

    // ---- TRANSFER + ENCODE ----
    try {
        itwSetStream(c.enc);
        const int mid = r0 + (r1 - r0 + 1) / 2;
        const int cut[3] = {r0, (r1 - r0 >= 2) ? mid : r1, r1};
        for (int s = 0; s < 2; s++) {
            const int a = cut[s], b = cut[s + 1];
            if (b <= a) continue;
            if (g.abort.load()) itw::fail_msg("stopped: another rank failed");
            const int64_t y0 = (int64_t)a * 4, y1 = (b == by) ? h : (int64_t)b * 4;
            rgba_surface sub = k.input;
            sub.height = (int)(y1 - y0);
            if (src_here) {
                sub.ptr = k.input.ptr + y0 * (int64_t)k.input.stride;
            } else {
                // host -> this GPU over its own PCIe link, or owner GPU -> this GPU over xGMI
                uint8_t* dpos = in + (size_t)(y0 - first_row) * pitch;
                upload_rows(dpos, pitch, k, y0, y1, row_bytes, c.enc);
                sub.ptr = dpos;
                sub.stride = (int32_t)pitch;
            }
            uint8_t* o = out + (size_t)(a - r0) * bx * k.bpb;
            const size_t nbytes = (size_t)(b - a) * bx * k.bpb;
            itwClearError();
            k.fn(&sub, o);                                                            // device pointers: asynchronous on c.enc
            if (const char* e = itwLastError()) itw::fail_msg("%s", e);
            if (!dst_here) {                                                          // (else: encoded in place)
                ITW_CHECK(hipEventRecord(c.ev[s], c.enc));
                ITW_CHECK(hipStreamWaitEvent(c.xfer, c.ev[s], 0));                    // the gather of this half runs under the next half's encode
                uint8_t* dpos = k.output + (size_t)a * bx * k.bpb;
                if (!k.dst_dev)      ITW_CHECK(hipMemcpyAsync(dpos, o, nbytes, hipMemcpyDeviceToHost, c.xfer));
                else if (k.use_rccl) ITW_NCCL(g.rccl.Send(o, nbytes, ncclUint8, k.dst_rank, g.comms[c.rank], c.xfer));
                else                 ITW_CHECK(hipMemcpyPeerAsync(dpos, k.dst_device, o, c.device, nbytes, c.xfer));
            }
            if (k.fail_rank == c.rank && k.fail_stage == 2) itw::fail_msg("injected failure after the first half-band (ITW_MULTIGPU_TEST_FAIL)");
        }
        // the rank that owns `output` posts the matching receives, one group per half so halves complete independently
        if (k.use_rccl && k.dst_dev && c.rank == k.dst_rank) {
            for (int s = 0; s < 2; s++) {
                if (g.abort.load()) itw::fail_msg("stopped: another rank failed");
                ITW_NCCL(g.rccl.GroupStart());
                for (int p = 0; p < k.ranks; p++) {
                    if (p == c.rank) continue;
                    int p0, p1;
                    band_rows(by, p, k.ranks, p0, p1);
                    if (p1 <= p0) continue;
                    const int pm = p0 + (p1 - p0 + 1) / 2;
                    const int pc[3] = {p0, (p1 - p0 >= 2) ? pm : p1, p1};
                    if (pc[s + 1] <= pc[s]) continue;
                    ITW_NCCL(g.rccl.Recv(k.output + (size_t)pc[s] * bx * k.bpb, (size_t)(pc[s + 1] - pc[s]) * bx * k.bpb, ncclUint8, p, g.comms[c.rank], c.xfer));
                }
                ITW_NCCL(g.rccl.GroupEnd());
            }
        }
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
// cpulist -> sched_setaffinity.  Anything missing (no sysfs, node -1, one-node box): the thread keeps its affinity.
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
    cpu_set_t set;
    CPU_ZERO(&set);
    int any = 0;
    for (const char* p = list; *p;) {                            // "0-31,64-95"
        char* q = nullptr;
        const long a = std::strtol(p, &q, 10);
        if (q == p) break;
        long b = a;
        if (*q == '-') { p = q + 1; b = std::strtol(p, &q, 10); }
        for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; cpu++) { CPU_SET((int)cpu, &set); any = 1; }
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

// RCCL is usable for a call when every rank sits on its own device; communicators are (re)built when the rank count changes
// or after a call that had to abort them
bool prepare_rccl(int ranks)
{
    const char* t = std::getenv("ITW_MULTIGPU_TRANSPORT");
    if (t && !std::strcmp(t, "peer")) return false;
    if (ranks > g.devices || !g.rccl.load()) return false;
    if (g.comms_aborted) { g.comms.clear(); g.comms_aborted = false; }        // ncclCommAbort released them already
    if ((int)g.comms.size() == ranks) return true;
    for (ncclComm_t c : g.comms) if (c) (void)g.rccl.CommDestroy(c);
    g.comms.assign((size_t)ranks, nullptr);
    std::vector<int> devs((size_t)ranks);
    for (int i = 0; i < ranks; i++) devs[(size_t)i] = i;
    if (g.rccl.CommInitAll(g.comms.data(), ranks, devs.data()) != ncclSuccess) { g.comms.clear(); return false; }
    return true;
}

} // namespace

extern "C" {

int itwMultiGpuRanks(void)
{
    const char* e = std::getenv("ITW_MULTIGPU_RANKS");
    const int n = e ? std::atoi(e) : 0;
    return n >= 1 ? (n > 64 ? 64 : n) : device_count();
}

const char* itwMultiGpuTransport(void) { return g.transport; }

int itwMultiGpuPeerLinks(void) { return g.peer_links.load(); }

bool itwCompressImageMultiGPU(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks)
{
    bool ok = false;
    itwClearError();
    itw::guarded([&] {
        if (!input || !input->ptr || !output || !cmpFunc) itw::fail_msg("itwCompressImageMultiGPU: null argument");
        std::lock_guard<std::mutex> one(g.submit);
        Call k;
        k.input = *input; k.output = output; k.fn = cmpFunc;
        k.keep_partial = dxgi_format == ITW_DXGI_FORMAT_BC4_UNORM || dxgi_format == ITW_DXGI_FORMAT_BC5_UNORM;
        k.bpb = GetBytesPerBlock(dxgi_format);
        k.texel_bytes = (dxgi_format == ITW_DXGI_FORMAT_BC6H_UF16 || dxgi_format == ITW_DXGI_FORMAT_BC6H_SF16) ? 8 : 4;
        const int by = k.keep_partial ? (input->height + 3) / 4 : input->height / 4;
        if (by <= 0 || input->width < (k.keep_partial ? 1 : 4)) { ok = true; return; }
        int n = ranks > 0 ? ranks : itwMultiGpuRanks();
        n = n > 64 ? 64 : (n > by ? by : n);
        k.ranks = n;
        ensure_ranks(n);
        k.src_device = device_of(input->ptr); k.src_dev = k.src_device >= 0;
        k.dst_device = device_of(output);     k.dst_dev = k.dst_device >= 0;
        k.dst_rank = k.dst_dev ? k.dst_device % g.devices : -1;                  // rank r lives on device r % devices: the lowest one there
        if (k.dst_dev && k.dst_rank >= n) k.dst_rank = -1;                        // no rank on the owner: peer copies only
        k.use_rccl = k.dst_dev && k.dst_rank >= 0 && n > 1 && prepare_rccl(n);
        g.transport = k.use_rccl ? "rccl" : "peer";
        if (const char* e = std::getenv("ITW_MULTIGPU_TEST_FAIL")) {              // "rank:stage" (tests of the failure model)
            int r = -1, st = 0;
            if (std::sscanf(e, "%d:%d", &r, &st) == 2) { k.fail_rank = r; k.fail_stage = st; }
        }
        {
            std::lock_guard<std::mutex> lk(g.m);
            g.call = k;
            g.ready = 0; g.prepare_failed = false; g.abort.store(false);
            for (int i = 0; i < n; i++) { g.ranks[(size_t)i]->pending = true; g.ranks[(size_t)i]->failed = false; }
            g.outstanding = n;
        }
        g.work.notify_all();
        std::unique_lock<std::mutex> lk(g.m);
        g.done.wait(lk, [&] { return g.outstanding == 0; });
        // report the rank that failed on its own, not one that merely stopped because of it
        int first = -1;
        for (int i = 0; i < n; i++)
            if (g.ranks[(size_t)i]->failed && (first < 0 || std::strstr(g.ranks[(size_t)first]->msg, "stopped: another rank failed"))) first = i;
        if (first >= 0) { itw::Failure f; std::snprintf(f.msg, sizeof f.msg, "%s", g.ranks[(size_t)first]->msg); lk.unlock(); throw f; }
        ok = true;
    });
    return ok;
}

} // extern "C"
