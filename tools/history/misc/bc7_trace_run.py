"""a few BC7 calls on one content kind, for rocprofv3 kernel traces: python tools/bc7_trace_run.py <profile> <I3|I3opaque|baboon>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces
prof, kind = sys.argv[1], sys.argv[2]
size = 4096
if kind.startswith("I3"):
    img = surfaces.ldr_smooth(size, size)
    if kind == "I3opaque":
        img[..., 3] = 255
else:
    z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
    img = np.ascontiguousarray(np.tile(z["baboon"], (16, 16, 1)))
d = torch.from_numpy(img).cuda()
out = torch.empty(size * size, dtype=torch.uint8, device="cuda")
for _ in range(4):
    itw_amd.compress("bc7", d, prof, out=out)
torch.cuda.synchronize()
