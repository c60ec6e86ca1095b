"""Device-resident BC7 call time by launch shape (deep / wide) and call size, to place ITW_BC7_WIDE_MAX_BLOCKS.
Runs on the GPU box.  Also checks that both shapes return the same bytes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0")
base = surfaces.ldr_smooth(4096, 4096)
profiles = sys.argv[1].split(",") if len(sys.argv) > 1 else ["slow", "basic", "alpha_basic", "veryfast", "alpha_slow"]
fmt = sys.argv[2] if len(sys.argv) > 2 else "bc7"
if fmt == "bc6h":
    base = surfaces.hdr_smooth(4096, 4096)
# optional third argument: content of the LDR surface -- I3 (default: smooth fields + noise, alpha field as generated), I3opaque, baboon (the
# reference's sample tiled, opaque): the bounded BC7 order (csrc/bc7.hip) makes the deep shape's time content dependent
content = sys.argv[3] if len(sys.argv) > 3 else "I3"
if fmt == "bc7" and content == "I3opaque":
    base = base.copy(); base[..., 3] = 255
if fmt == "bc7" and content == "baboon":
    z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
    base = np.ascontiguousarray(np.tile(z["baboon"], (16, 16, 1)))
    # rows of the probe cut through whole tiles: every call size sees the same mix of blocks
print("content:", content)
print(f"{'profile':<12} {'rows x 4096':>12} {'blocks':>9} {'deep ms':>9} {'wide ms':>9} {'wide Mpix/s':>12} same")
for prof in profiles:
    for rows in (8, 32, 64, 128, 256, 384, 512, 1024, 4096):
        img = torch.from_numpy(np.ascontiguousarray(base[:rows])).to(dev)
        n = rows // 4 * 1024
        out = {}
        ms = {}
        for path in ("deep", "wide"):
            itw_amd.set_bc7_path(path)
            o = itw_amd.compress(fmt, img, prof)
            torch.cuda.synchronize()
            reps = 5 if rows >= 1024 else 20
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                itw_amd.compress(fmt, img, prof, out=o)
            b.record()
            torch.cuda.synchronize()
            ms[path] = a.elapsed_time(b) / reps
            out[path] = o.cpu().numpy()
        same = bool((out["deep"] == out["wide"]).all())
        print(f"{prof:<12} {rows:>12} {n:>9} {ms['deep']:>9.3f} {ms['wide']:>9.3f} {rows * 4096 / ms['wide'] / 1e3:>12.1f} {same}", flush=True)
itw_amd.set_bc7_path("auto")
