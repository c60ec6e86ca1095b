"""Does running a refinement kernel (2 waves/SIMD, 251 VGPRs) beside a scan kernel (4 waves/SIMD, 128 VGPRs) beat running them
one after the other?  Two host threads (each with its own BC7 workspace) encode complementary parts of one 4096^2 surface on two
streams; compare with one call for the whole surface.  Runs on the GPU box."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0")
img = torch.from_numpy(surfaces.ldr_smooth(4096, 4096)).to(dev)
out = torch.empty(4096 * 4096, dtype=torch.uint8, device=dev)
itw_amd.set_bc7_path("deep")
prof = sys.argv[1] if len(sys.argv) > 1 else "slow"

def whole():
    itw_amd.compress("bc7", img, prof, out=out)

def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

print("one call            %.3f ms" % timed(whole))
for cuts in ([1024], [2048], [512, 2304], [1024, 2048, 3072], [512, 1024, 2048, 3072]):
    bounds = [0] + cuts + [4096]
    parts = [(bounds[i], bounds[i + 1]) for i in range(len(bounds) - 1)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    def worker(k):
        with torch.cuda.stream(streams[k]):
            for i, (y0, y1) in enumerate(parts):
                if i % 2 == k:
                    itw_amd.compress("bc7", img[y0:y1], prof, out=out[y0 * 1024 * 4: y1 * 1024 * 4])   # 4096/4 blocks per row * 16 B / 4 rows
    def split():
        th = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
        for t in th: t.start()
        for t in th: t.join()
    print("rows %-28s two threads / streams  %.3f ms" % (parts, timed(split)))
ref = out.clone(); whole(); torch.cuda.synchronize()
print("same bytes:", bool((ref == out).all()))
