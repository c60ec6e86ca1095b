"""GPU box: actual shader clock under the BC1 / BC3 kernels and the BC7 `slow` scan (gpurun_variants/lib_clockprobe.so copied over the
product library first)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces
L = itw_amd.lib()
dev = torch.device("cuda:0")
img = torch.from_numpy(surfaces.ldr_smooth(4096, 4096)).to(dev)
big = img.repeat(4, 4, 1).contiguous()
out = torch.empty(4096 * 4096, dtype=torch.uint8, device=dev)
outb = torch.empty(16384 * 16384, dtype=torch.uint8, device=dev)
buf = np.zeros(4, dtype=np.uint64)
def run(name, fn, reader, reps):
    fn(); torch.cuda.synchronize()
    getattr(L, reader)(buf.ctypes.data_as(C.c_void_p))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    getattr(L, reader)(buf.ctypes.data_as(C.c_void_p))
    cyc, ticks, n = int(buf[0]), int(buf[1]), int(buf[2])
    print(f"{name:28s} {a.elapsed_time(b) / reps * 1e3:9.1f} us per call | workgroup lifetime {cyc / n:10.0f} shader cycles = {ticks / n * 10:9.0f} ns "
          f"-> {cyc / (ticks * 10.0):5.2f} GHz  ({n // reps} workgroups per call)")
run("bc1 4096^2 x200", lambda: itw_amd.compress("bc1", img, out=out), "itwProbeReadClockBc13", 200)
run("bc3 4096^2 x200", lambda: itw_amd.compress("bc3", img, out=out), "itwProbeReadClockBc13", 200)
run("bc1 16384^2 x20", lambda: itw_amd.compress("bc1", big, out=outb), "itwProbeReadClockBc13", 20)
run("bc7 slow scan 4096^2 x5", lambda: itw_amd.compress("bc7", img, "slow", out=out), "itwProbeReadClockBc7", 5)
