"""BC7 `slow` / `alpha_slow` at 4096^2 on content of different kinds (HIP events, device resident): what the bounded mode order (modes 1/3
last, only where their exact lower bound is below the other modes' result; csrc/bc7.hip) costs or saves.  Run once per setting of
ITW_BC7_BOUND (read once per process): tools/gpu_bounded.sh.  Also prints the share of blocks that still visit modes 1/3."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
size = 4096
z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz")); z2 = np.load(os.path.join(ROOT, "tests", "golden", "samples2.npz"))

def tiled(a):
    a = a[:a.shape[0] // 4 * 4, :a.shape[1] // 4 * 4]
    ry, rx = -(-size // a.shape[0]), -(-size // a.shape[1])
    return np.ascontiguousarray(np.tile(a, (ry, rx, 1))[:size, :size])

contents = [("I3 ldr_smooth", surfaces.ldr_smooth(size, size)), ("I2 colors16m", surfaces.colors_16m(size)),
            ("baboon tiled", tiled(z["baboon"])), ("monkey tiled", tiled(z["monkey"])), ("normals tiled", tiled(z2["normals"])),
            ("landscape tiled", tiled(z2["landscape_detail"])), ("test_a tiled", tiled(z2["test_a"]))]
if os.environ.get("BOUNDED_QUICK"):
    contents = [c for c in contents if c[0].split()[0] in ("I3", "I2", "baboon", "test_a")]
out = torch.empty(size * size, dtype=torch.uint8, device=dev)

def t(img, prof, n=4):
    itw_amd.compress("bc7", img, prof, out=out); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): itw_amd.compress("bc7", img, prof, out=out)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n

print("ITW_BC7_BOUND =", os.environ.get("ITW_BC7_BOUND", "(default: on)"))
for name, img in contents:
    if img is None:
        continue
    d = torch.from_numpy(img).to(dev)
    bounds = itw_amd.bc7_two_subset_bounds(d[:1024]).min(dim=1).values
    op = d.clone(); op[..., 3] = 255
    print(f"{name:16s} slow {t(d, 'slow'):7.3f} ms   alpha_slow (as is) {t(d, 'alpha_slow'):7.3f} ms   alpha_slow (opaque) {t(op, 'alpha_slow'):7.3f} ms"
          f"   median of the smallest two-subset bound {bounds.median().item():8.1f}", flush=True)
