// round 6 probe: what does pinning the caller's pageable surface for one call cost, against the pageable copy it would replace?
// hipcc --offload-arch=gfx950 -O2 host_register_probe.hip -o host_register_probe && ./host_register_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static double ms(std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); }
int main()
{
    const size_t n = 64u << 20;
    void* h = nullptr;
    if (posix_memalign(&h, 4096, n)) return 1;
    memset(h, 1, n);
    void* d = nullptr;
    hipMalloc(&d, n);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 3; rep++) {
        auto t = std::chrono::steady_clock::now();
        hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        const double pageable = ms(t);
        t = std::chrono::steady_clock::now();
        hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault);
        const double reg = ms(t);
        t = std::chrono::steady_clock::now();
        hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s); hipStreamSynchronize(s);
        const double pinned = ms(t);
        t = std::chrono::steady_clock::now();
        hipHostUnregister(h);
        const double unreg = ms(t);
        printf("64 MiB H2D: pageable %.3f ms (%.1f GB/s) | hipHostRegister %.3f ms (%s) + pinned copy %.3f ms (%.1f GB/s) + hipHostUnregister %.3f ms\n",
               pageable, n / pageable / 1e6, reg, hipGetErrorString(e), pinned, n / pinned / 1e6, unreg);
    }
    return 0;
}
