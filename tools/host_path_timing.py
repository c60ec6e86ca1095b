"""End-to-end rate of the C ABI with HOST pointers (what the Photoshop plugin passes): H2D + kernel + D2H inside one
synchronous call, pageable numpy memory.  Reported next to the device-resident rate in DESIGN.md; never bench.py's value."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
torch.cuda.set_device(0)
for fmt, prof in (("bc1", None), ("bc3", None), ("bc7", "basic"), ("bc7", "slow"), ("bc6h", "fast"), ("bc6h", "slow")):
    img = surfaces.hdr_smooth(size, size) if fmt == "bc6h" else surfaces.ldr_smooth(size, size)
    itw_amd.compress_numpy(fmt, img, prof)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); itw_amd.compress_numpy(fmt, img, prof); ts.append(time.perf_counter() - t0)
    d = torch.from_numpy(img).cuda(); out = torch.empty(size * size // 16 * itw_amd.BYTES_PER_BLOCK[fmt], dtype=torch.uint8, device="cuda")
    itw_amd.compress(fmt, d, prof, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter(); itw_amd.compress(fmt, d, prof, out=out); torch.cuda.synchronize(); tk = time.perf_counter() - t0
    print(f"{fmt:5s} {prof or '-':6s} host-pointer call {min(ts)*1e3:8.2f} ms = {size*size/min(ts)/1e6:9.1f} Mpix/s   device-resident {tk*1e3:8.3f} ms", flush=True)
