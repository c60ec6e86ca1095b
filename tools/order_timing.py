"""Round 5: BC7 `slow` (and `alpha_slow`) at 4096^2 by content and by mode-order policy, device resident (HIP events) and through host
pointers (wall clock of the synchronous call).  One process per environment setting (the knobs are read once): tools/evidence.sh.
  argv: list of content names (default: I3 I2 baboon test_a)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
size = int(os.environ.get("ORDER_SIZE", "4096"))
z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz")); z2 = np.load(os.path.join(ROOT, "tests", "golden", "samples2.npz"))

def tiled(a):
    a = a[:a.shape[0] // 4 * 4, :a.shape[1] // 4 * 4]
    ry, rx = -(-size // a.shape[0]), -(-size // a.shape[1])
    return np.ascontiguousarray(np.tile(a, (ry, rx, 1))[:size, :size])

def half_half():
    a = surfaces.ldr_smooth(size, size).copy()
    a[size // 2:] = tiled(z["baboon"])[size // 2:]
    return a

makers = {"I3": lambda: surfaces.ldr_smooth(size, size), "I2": lambda: surfaces.colors_16m(size), "baboon": lambda: tiled(z["baboon"]),
          "monkey": lambda: tiled(z["monkey"]), "test_a": lambda: tiled(z2["test_a"]), "landscape": lambda: tiled(z2["landscape_detail"]),
          "mixed": half_half}
names = sys.argv[1:] or ["I3", "I2", "baboon", "test_a"]
profs = os.environ.get("ORDER_PROFILES", "slow").split(",")
host = os.environ.get("ORDER_HOST", "1") == "1"
out = torch.empty(size * size, dtype=torch.uint8, device=dev)

def t_dev(img, prof, n=6):
    itw_amd.compress("bc7", img, prof, out=out); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(n): itw_amd.compress("bc7", img, prof, out=out)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / n)
    return best

def t_host(img, prof, n=5):
    itw_amd.compress_numpy("bc7", img, prof)
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); itw_amd.compress_numpy("bc7", img, prof); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3

tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("ITW_"))
print(f"== {tag or '(defaults)'}", flush=True)
for name in names:
    img = makers[name]()
    d = torch.from_numpy(img).to(dev)
    line = f"{name:10s}"
    for prof in profs:
        src = d
        if prof == "alpha_slow":
            src = d.clone(); src[..., 3] = 255
        line += f"  {prof} device {t_dev(src, prof):7.3f} ms"
        if host:
            h = img if prof != "alpha_slow" else np.ascontiguousarray(np.concatenate([img[..., :3], np.full_like(img[..., :1], 255)], axis=2))
            line += f"  host-pointer call {t_host(h, prof):7.3f} ms"
    print(line, flush=True)
