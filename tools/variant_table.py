"""Kernel time (device-resident, HIP events) and a SHA-256 of the emitted blocks for the workloads a kernel variant is judged on: the
presets the plugin selects, the slow presets, on the bench surface and on the reference's baboon.png tiled.  tools/gpu_variants.sh runs it
once per gpurun_variants/lib_*.so.  Usage: python tools/variant_table.py [filter substring]"""
import hashlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
size = 4096
flt = sys.argv[1] if len(sys.argv) > 1 else ""
ldr_h = surfaces.ldr_smooth(size, size)
z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
reps = -(-size // z["baboon"].shape[0])
nat_h = np.ascontiguousarray(np.tile(z["baboon"], (reps, reps, 1))[:size, :size])
surf = {"synthetic": torch.from_numpy(ldr_h).to(dev), "opaque": torch.from_numpy(surfaces.ldr_alpha_variant(ldr_h, "opaque")).to(dev),
        "baboon": torch.from_numpy(nat_h).to(dev), "hdr": torch.from_numpy(surfaces.hdr_smooth(size, size).view(np.int16)).to(dev)}
out = torch.empty(size * size, dtype=torch.uint8, device=dev)

def t(fmt, img, prof, n):
    itw_amd.compress(fmt, img, prof, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): itw_amd.compress(fmt, img, prof, out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

rows = [("bc7", "veryfast", "synthetic"), ("bc7", "basic", "synthetic"), ("bc7", "basic", "baboon"), ("bc7", "alpha_veryfast", "synthetic"), ("bc7", "alpha_basic", "synthetic"),
        ("bc7", "alpha_basic", "opaque"), ("bc7", "slow", "synthetic"), ("bc7", "slow", "baboon"), ("bc7", "alpha_slow", "synthetic"), ("bc7", "alpha_slow", "opaque"),
        ("bc7", "alpha_slow", "baboon"), ("bc6h", "fast", "hdr"), ("bc6h", "slow", "hdr"), ("bc1", None, "synthetic"), ("bc3", None, "synthetic")]
for fmt, prof, content in rows:
    tag = f"{fmt}_{prof or '-'}@{content}"
    if flt and flt not in tag:
        continue
    n = 50 if fmt in ("bc1", "bc3") else 5
    ms = min(t(fmt, surf[content], prof, n) for _ in range(2))
    nbytes = (size // 4) ** 2 * itw_amd.BYTES_PER_BLOCK[fmt]
    sha = hashlib.sha256(out[:nbytes].cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{tag:34s} {ms:9.4f} ms  {size * size / ms / 1e3:10.0f} Mpix/s  sha {sha}", flush=True)
