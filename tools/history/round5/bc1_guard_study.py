"""VERDICT r04 item 4 asked whether BC1 / BC3's covariance (compute_covar_dc_ugly, kernel.ispc:377-417: six sums of 16 exact products in texel
order) can come from integer moments (256 sum(xy) - 16 sum(x) sum(y) on v_dot4_u32_u8) under a guard that makes the float sums order-free.
This script measures how often the guard holds -- every partial sum of every one of the six accumulations stays below 2^16 in magnitude
(products are multiples of 1/256, so such sums are exact in fp32 whatever the order) -- on the bench surface and on natural images."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
from itw_amd import surfaces

def rate(img):
    h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
    t = img[:h, :w, :3].astype(np.float64).reshape(h // 4, 4, w // 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 3)
    c = t - t.mean(axis=1, keepdims=True)
    ok = np.ones(t.shape[0], bool)
    for a, b in ((0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)):
        part = np.cumsum(c[:, :, a] * c[:, :, b], axis=1)
        ok &= (np.abs(part) < 65536).all(axis=1)
    return ok.mean()

z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz")); z2 = np.load(os.path.join(ROOT, "tests", "golden", "samples2.npz"))
for name, img in (("I3 ldr_smooth 1024^2", surfaces.ldr_smooth(1024, 1024)), ("I3u uniform bytes", surfaces.ldr_uniform(512, 512)), ("baboon", z["baboon"]),
                  ("monkey", z["monkey"]), ("colors260k", z2["colors260k"]), ("test_a", z2["test_a"]), ("landscape", z2["landscape_detail"])):
    print(f"{name:22s} blocks whose six accumulations stay exact in any order: {100 * rate(img):6.2f} %")
