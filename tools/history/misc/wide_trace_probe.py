"""One BC7 call of `rows` x 4096 texels on the wide path, a few repetitions, for rocprofv3 --kernel-trace."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prof = sys.argv[2] if len(sys.argv) > 2 else "slow"
path = sys.argv[3] if len(sys.argv) > 3 else "wide"
fmt = sys.argv[4] if len(sys.argv) > 4 else "bc7"
dev = torch.device("cuda:0")
base = surfaces.hdr_smooth(rows, 4096) if fmt == "bc6h" else surfaces.ldr_smooth(rows, 4096)
img = torch.from_numpy(base).to(dev)
itw_amd.set_bc7_path(path)
o = itw_amd.compress(fmt, img, prof)
torch.cuda.synchronize()
for _ in range(10):
    itw_amd.compress(fmt, img, prof, out=o)
torch.cuda.synchronize()
