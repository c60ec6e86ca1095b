"""GPU box: prune rates from gpurun_variants/lib_pruneprobe.so (tools/variants/make_bc7_prune_probe.py).  The variant library is
loaded INSTEAD of the product (copy it over intel-texture-works-plugin_amd/lib/libispc_texcomp.so first: tools/round4/gpu_r04c.sh)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces

L = itw_amd.lib()
L.itwProbeReadCounters.argtypes = [C.c_void_p]
dev = torch.device("cuda:0")
itw_amd.set_bc7_path("deep")
cnt = np.zeros(8, dtype=np.uint32)
L.itwProbeReadCounters(cnt.ctypes.data)
gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz")))
inputs = [("I3 ldr_smooth 2048^2", surfaces.ldr_smooth(2048, 2048)), ("I2 colors-16M 4096^2", surfaces.colors_16m()),
          ("I1 baboon 512^2", gold["baboon"]), ("monkey (LDR photo)", gold["monkey"])]
for name, img in inputs:
    itw_amd.compress("bc7", torch.from_numpy(np.ascontiguousarray(img)).to(dev), "slow")
    torch.cuda.synchronize()
    L.itwProbeReadCounters(cnt.ctypes.data)
    c = cnt.astype(np.float64)
    print(f"{name:24s} modes 1/3 after 1 of 2 subsets: waves where ALL 64 blocks are already above their best {c[1] / max(c[0], 1):7.2%} "
          f"(single blocks {c[2] / max(64 * c[0], 1):6.2%}) | modes 0/2 after 2 of 3 subsets: {c[4] / max(c[3], 1):7.2%} (single blocks {c[5] / max(64 * c[3], 1):6.2%})")
