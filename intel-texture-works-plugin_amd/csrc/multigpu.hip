// multigpu.hip -- itwCompressImageMultiGPU (include/itw_multigpu.h): one surface over all GPUs of the node from ONE
// process, host code in C++: band per rank (win32Threads.cpp:217-231 on block rows), scatter of a device-resident
// surface by peer copies, gather of the output bands to the owner of `output` by RCCL send/recv (or peer copies), the
// gather of a rank's first half-band overlapping the encode of its second.
//
// One persistent host thread per rank, bound to its device: CompressBlocks* keeps per-thread, per-device state (stream,
// BC7 workspace), so a rank's thread is the natural owner of its streams and staging buffers.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>          // types and prototypes only: the symbols are resolved with dlsym on first use
#include <dlfcn.h>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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
        ITW_SYM(CommInitAll) ITW_SYM(CommDestroy) ITW_SYM(Send) ITW_SYM(Recv) ITW_SYM(GroupStart) ITW_SYM(GroupEnd) ITW_SYM(GetErrorString)
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
};

struct RankCtx {
    int rank = 0, device = 0;
    std::thread th;
    hipStream_t enc = nullptr, xfer = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    void* d_in = nullptr;  size_t in_cap = 0;
    void* d_out = nullptr; size_t out_cap = 0;
    bool pending = false, failed = false;
    char msg[384] = {0};
};

struct Group {
    std::mutex m, submit;
    std::condition_variable work, done;
    std::vector<RankCtx*> ranks;
    int devices = 1, outstanding = 0;
    bool quit = false;
    Call call;
    Rccl rccl;
    std::vector<ncclComm_t> comms;      // one per rank when RCCL is in use (ranks == distinct devices)
    const char* transport = "peer";
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

void run_rank(RankCtx& c, const Call& k)
{
    const int w = k.input.width, h = k.input.height;
    const int bx = k.keep_partial ? (w + 3) / 4 : w / 4, by = k.keep_partial ? (h + 3) / 4 : h / 4;
    int r0, r1;
    band_rows(by, c.rank, k.ranks, r0, r1);
    if (r1 <= r0) return;
    if (!c.enc) {
        ITW_CHECK(hipStreamCreateWithFlags(&c.enc, hipStreamNonBlocking));
        ITW_CHECK(hipStreamCreateWithFlags(&c.xfer, hipStreamNonBlocking));
        for (auto& e : c.ev) ITW_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    itwSetStream(c.enc);
    const size_t row_bytes = (size_t)w * k.texel_bytes;
    const size_t pitch = (row_bytes + 15) & ~(size_t)15;
    const bool src_here = k.src_dev && k.src_device == c.device;
    const bool dst_here = k.dst_dev && k.dst_device == c.device;
    const size_t band_out = (size_t)(r1 - r0) * bx * k.bpb;
    const int64_t first_row = (int64_t)r0 * 4;
    const int64_t last_row = (r1 == by) ? h : (int64_t)r1 * 4;                    // the last band keeps a partial block row (BC4/BC5)
    uint8_t* in = src_here ? nullptr : (uint8_t*)grow(c.d_in, c.in_cap, pitch * (size_t)(last_row - first_row));
    uint8_t* out = dst_here ? k.output + (size_t)r0 * bx * k.bpb : (uint8_t*)grow(c.d_out, c.out_cap, band_out);

    const int mid = r0 + (r1 - r0 + 1) / 2;
    const int cut[3] = {r0, (r1 - r0 >= 2) ? mid : r1, r1};
    for (int s = 0; s < 2; s++) {
        const int a = cut[s], b = cut[s + 1];
        if (b <= a) continue;
        const int64_t y0 = (int64_t)a * 4, y1 = (b == by) ? h : (int64_t)b * 4;
        rgba_surface sub = k.input;
        sub.height = (int)(y1 - y0);
        if (src_here) {
            sub.ptr = k.input.ptr + y0 * k.input.stride;
        } else {
            // host -> this GPU over its own PCIe link, or owner GPU -> this GPU over xGMI (hipMemcpyDefault: peer copy)
            uint8_t* dpos = in + (size_t)(y0 - first_row) * pitch;
            ITW_CHECK(hipMemcpy2DAsync(dpos, pitch, k.input.ptr + y0 * k.input.stride, (size_t)k.input.stride, row_bytes, (size_t)(y1 - y0),
                                       k.src_dev ? hipMemcpyDefault : hipMemcpyHostToDevice, c.enc));
            sub.ptr = dpos;
            sub.stride = (int32_t)pitch;
        }
        uint8_t* o = out + (size_t)(a - r0) * bx * k.bpb;
        const size_t nbytes = (size_t)(b - a) * bx * k.bpb;
        itwClearError();
        k.fn(&sub, o);                                                            // device pointers: asynchronous on c.enc
        if (const char* e = itwLastError()) itw::fail_msg("%s", e);
        if (dst_here) continue;                                                   // encoded in place
        ITW_CHECK(hipEventRecord(c.ev[s], c.enc));
        ITW_CHECK(hipStreamWaitEvent(c.xfer, c.ev[s], 0));                        // the gather of this half runs under the next half's encode
        uint8_t* dpos = k.output + (size_t)a * bx * k.bpb;
        if (!k.dst_dev)      ITW_CHECK(hipMemcpyAsync(dpos, o, nbytes, hipMemcpyDeviceToHost, c.xfer));
        else if (k.use_rccl) ITW_NCCL(g.rccl.Send(o, nbytes, ncclUint8, k.dst_rank, g.comms[c.rank], c.xfer));
        else                 ITW_CHECK(hipMemcpyPeerAsync(dpos, k.dst_device, o, c.device, nbytes, c.xfer));
    }
    // the rank that owns `output` posts the matching receives, one group per half so halves complete independently
    if (k.use_rccl && k.dst_dev && c.rank == k.dst_rank) {
        for (int s = 0; s < 2; s++) {
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
}

void rank_main(RankCtx* c)
{
    (void)hipSetDevice(c->device);
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
        lk.lock();
        c->pending = false;
        c->failed = bad;
        if (bad) std::snprintf(c->msg, sizeof c->msg, "rank %d (device %d): %s", c->rank, c->device, fail.msg);
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
bool prepare_rccl(int ranks)
{
    const char* t = std::getenv("ITW_MULTIGPU_TRANSPORT");
    if (t && !std::strcmp(t, "peer")) return false;
    if (ranks > g.devices || !g.rccl.load()) return false;
    if ((int)g.comms.size() == ranks) return true;
    for (ncclComm_t c : g.comms) (void)g.rccl.CommDestroy(c);
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
        {
            std::lock_guard<std::mutex> lk(g.m);
            g.call = k;
            for (int i = 0; i < n; i++) { g.ranks[(size_t)i]->pending = true; g.ranks[(size_t)i]->failed = false; }
            g.outstanding = n;
        }
        g.work.notify_all();
        std::unique_lock<std::mutex> lk(g.m);
        g.done.wait(lk, [&] { return g.outstanding == 0; });
        for (int i = 0; i < n; i++)
            if (g.ranks[(size_t)i]->failed) { itw::Failure f; std::snprintf(f.msg, sizeof f.msg, "%s", g.ranks[(size_t)i]->msg); lk.unlock(); throw f; }
        ok = true;
    });
    return ok;
}

} // extern "C"
