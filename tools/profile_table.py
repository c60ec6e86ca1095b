"""Kernel time of every quality preset at 4096^2 (device-resident, HIP events): the table in DESIGN.md section 3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
size = 4096
ldr = torch.from_numpy(surfaces.ldr_smooth(size, size)).to(dev)
hdr = torch.from_numpy(surfaces.hdr_smooth(size, size).view(np.int16)).to(dev)
out = torch.empty(size * size, dtype=torch.uint8, device=dev)

def t(fmt, img, prof, n):
    itw_amd.compress(fmt, img, prof, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): itw_amd.compress(fmt, img, prof, out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

rows = [("bc1", ldr, None, 50), ("bc3", ldr, None, 50), ("bc4", ldr, None, 50), ("bc5", ldr, None, 50)]
rows += [("bc7", ldr, p, 5) for p in itw_amd.BC7_PROFILES] + [("bc6h", hdr, p, 5) for p in itw_amd.BC6H_PROFILES]
for fmt, img, prof, n in rows:
    ms = t(fmt, img, prof, n)
    print(f"{fmt:5s} {prof or '-':16s} {ms:9.4f} ms  {size * size / ms / 1e3:12.0f} Mpix/s")
