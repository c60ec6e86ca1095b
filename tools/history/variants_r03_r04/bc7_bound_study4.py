"""Feasibility study 4 (CPU, next round): the REVERSED order -- modes 1,3,4,5,6 first, then modes 0/2 only where some three-subset shape's
lower bound (same construction as the two-subset bound: sum over 3 subsets of (sqrt(R) - sqrt(3)/2 sqrt(n))_+^2) is below that result.
Would natural content, where modes 1/3 win most blocks, skip the three-subset scans?"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import pyoracle
from itw_amd import surfaces
from bc7_bound_study2 import subsets
from bc7_bound_study3 import min_lb, errs

def study(name, img):
    h, w = img.shape[:2]
    tex = img[..., :3].astype(np.float64).reshape(h // 4, 4, w // 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 3)
    nb = tex.shape[0]
    planar = np.zeros((nb, 64), dtype=np.float32)
    planar[:, :48] = tex.transpose(0, 2, 1).reshape(nb, 48); planar[:, 48:] = 255
    sub = subsets()
    full = pyoracle.bc7_profile("slow")
    no02 = pyoracle.bc7_profile("slow"); no02.mode_selection[0] = 0
    e_full, m_full = errs(planar, full)
    e_b, _ = errs(planar, no02)
    lb2 = min_lb(tex, sub, range(64, 128), 3, 1, np.sqrt(3) / 2)          # mode 2: all 64 three-subset shapes
    lb0 = min_lb(tex, sub, range(64, 80), 3, 1, np.sqrt(3) / 2)           # mode 0: the first 16
    need2 = 0.999 * lb2 - 1 <= e_b
    need0 = 0.999 * lb0 - 1 <= e_b
    won02 = (m_full == 0) | (m_full == 2)
    print(f"{name:12s} blocks {nb}: modes 0/2 win {100 * won02.mean():5.1f} %; the reversed order still visits mode 2 for {100 * need2.mean():5.1f} %, mode 0 for "
          f"{100 * need0.mean():5.1f} % of the blocks (wrongly dropped: {int((won02 & ~(need2 | need0)).sum())})", flush=True)

if __name__ == "__main__":
    g = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(g, "inputs.npz")); z2 = np.load(os.path.join(g, "samples2.npz"))
    study("I3 smooth", surfaces.ldr_smooth(4096, 4096)[1024:1280, 2048:2304])
    for nm, a in (("baboon", z["baboon"]), ("monkey", z["monkey"][:216, :216]), ("colors260k", z2["colors260k"][:256, :256]),
                  ("normals", z2["normals"]), ("test_a", z2["test_a"][:256, :256]), ("landscape", z2["landscape_detail"][:336, :124])):
        study(nm, np.ascontiguousarray(a))
