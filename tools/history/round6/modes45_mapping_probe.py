"""Round 6 study: modes 4/5 (and 6) alone at 4096^2 under the two mappings the tree holds --
deep  = lane = block, the lane walks its 9 (RGB) / 12 (RGBA) candidates one after the other (bc7_finish_all's modes_45);
wide  = one (rotation, candidate) per TASK: every wave runs one candidate of 64 blocks, ordered argmin over the tasks afterwards
        (bc7_wide_phase2<SINGLES> + bc7_wide_commit) -- the "lanes = block x candidate, wave-uniform candidate" mapping.
Same bytes (checked)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
size = 4096
z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
reps = -(-size // z["baboon"].shape[0])
imgs = {"synthetic": torch.from_numpy(surfaces.ldr_smooth(size, size)).to(dev),
        "baboon": torch.from_numpy(np.ascontiguousarray(np.tile(z["baboon"], (reps, reps, 1))[:size, :size])).to(dev)}
out = torch.empty(size * size, dtype=torch.uint8, device=dev)
ref = torch.empty_like(out)

def t(img, s, n=3):
    itw_amd.compress("bc7", img, s, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): itw_amd.compress("bc7", img, s, out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for content, img in imgs.items():
    for prof in ("slow", "basic", "alpha_slow", "alpha_basic"):
        for name, keep in (("modes 4/5", (2,)), ("mode 6", (3,)), ("modes 4/5/6", (2, 3))):
            s = itw_amd.bc7_profile(prof)
            for i in range(4): s.mode_selection[i] = i in keep
            row = []
            for path in ("deep", "wide"):
                itw_amd.set_bc7_path(path)
                ms = t(img, s)
                if path == "deep": ref.copy_(out)
                row.append(ms)
            same = bool(torch.equal(ref, out))
            print(f"{content:10s} {prof:12s} only {name:12s} deep {row[0]:7.3f} ms   wide {row[1]:7.3f} ms   same bytes {same}", flush=True)
itw_amd.set_bc7_path("auto")
