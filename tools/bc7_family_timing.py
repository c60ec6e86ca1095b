"""Times the BC7 kernel with one mode family enabled at a time (custom settings), on the bench surface."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
img = torch.from_numpy(surfaces.ldr_smooth(size, size)).to(dev)
out = torch.empty(size * size, dtype=torch.uint8, device=dev)

def t(s, n=3):
    itw_amd.compress("bc7", img, s, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): itw_amd.compress("bc7", img, s, out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

for prof in ("slow", "alpha_slow", "basic"):
    base = itw_amd.bc7_profile(prof)
    print(f"{prof}: all {t(base):8.2f} ms")
    for name, keep in (("modes 0/2", 0), ("modes 1/3/7", 1), ("modes 4/5", 2), ("mode 6", 3)):
        s = itw_amd.bc7_profile(prof)
        for i in range(4): s.mode_selection[i] = (i == keep) and bool(base.mode_selection[i])
        print(f"   only {name:12s} {t(s):8.2f} ms")
    if prof == "slow":
        s = itw_amd.bc7_profile(prof)
        for i in range(4): s.mode_selection[i] = (i == 0)
        s.skip_mode2 = True
        print(f"   only mode 0 (16 shapes) {t(s):8.2f} ms")
        s = itw_amd.bc7_profile(prof)
        for i in range(4): s.mode_selection[i] = (i == 1)
        s.fastSkipTreshold_mode3 = 0
        print(f"   only mode 1 (64)        {t(s):8.2f} ms")
        s.fastSkipTreshold_mode1 = 1; s.fastSkipTreshold_mode3 = 0
        print(f"   ranking + 1 shape mode1 {t(s):8.2f} ms")
