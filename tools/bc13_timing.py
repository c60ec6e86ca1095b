"""BC1 / BC3 kernel time at 4096^2 and 16384^2 (HIP events around back-to-back launches); used by tools/evidence.sh and the grid sweeps (tools/build_variant.sh + tools/gpu_variants.sh)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
base = torch.from_numpy(surfaces.ldr_smooth(4096, 4096)).to(dev)
for size in (4096, 16384):
    img = base if size == 4096 else base.repeat(4, 4, 1).contiguous()
    out = torch.empty(size * size, dtype=torch.uint8, device=dev)
    for fmt in ("bc1", "bc3"):
        n = 200 if size == 4096 else 40
        for _ in range(10): itw_amd.compress(fmt, img, None, out=out)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n): itw_amd.compress(fmt, img, None, out=out)
            b.record(); torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / n)
        bytes_ = (size // 4) ** 2 * (72 if fmt == "bc1" else 80)
        print(f"{fmt} {size:6d}  {best * 1e3:9.2f} us   {bytes_ / best / 1e6:8.1f} GB/s  = {bytes_ / best / 1e6 / 8000:.3f} of 8 TB/s", flush=True)
