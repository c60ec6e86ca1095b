"""Feasibility study 3 (CPU): a TWO-PHASE order for RGB profiles.  Phase A = modes {0,2} + {4,5,6} -> incumbent G per block; phase B = modes
{1,3} only for the blocks where some two-subset shape's exact lower bound is <= G (any mode-1/3 encoding of shape p costs >= LB(p):
its palette is rounded points on a segment per subset).  Reports the share of blocks phase B still has to visit."""
import os, sys, re, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
from oracle import pyoracle
from itw_amd import surfaces
from bc7_bound_study2 import subsets, lam_upper

def min_lb(tex, sub, shapes, nsub, how, delta):
    nb = tex.shape[0]
    best = np.full(nb, np.inf)
    for p in shapes:
        lb = np.zeros(nb)
        for s in range(nsub):
            m = sub[p] == s
            n = m.sum()
            x = tex[:, m, :]
            x = x - x.mean(axis=1, keepdims=True)
            c = np.einsum("bki,bkj->bij", x, x)
            tr = np.trace(c, axis1=1, axis2=2)
            lam = np.linalg.eigvalsh(c)[:, -1] if how == "exact" else lam_upper(c, how)
            r = np.maximum(tr - lam, 0)
            lb += np.maximum(np.sqrt(r) - delta * np.sqrt(n), 0) ** 2
        best = np.minimum(best, lb)
    return best

def errs(planar, settings):
    L = pyoracle.lib()
    fn = L.oracle_bc7_block
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]; fn.restype = None
    out = np.zeros(planar.shape[0], dtype=np.float32)
    data = (C.c_uint32 * 4)(); e = C.c_float()
    modes = np.zeros(planar.shape[0], dtype=np.int32)
    for b in range(planar.shape[0]):
        fn(planar[b].ctypes.data, C.byref(settings), data, C.byref(e))
        out[b] = e.value
        modes[b] = (data[0] & -data[0]).bit_length() - 1
    return out, modes

def study(name, img):
    h, w = img.shape[:2]
    tex = img[..., :3].astype(np.float64).reshape(h // 4, 4, w // 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 3)
    nb = tex.shape[0]
    planar = np.zeros((nb, 64), dtype=np.float32)
    planar[:, :48] = tex.transpose(0, 2, 1).reshape(nb, 48); planar[:, 48:] = 255
    sub = subsets()
    full = pyoracle.bc7_profile("slow")
    a = pyoracle.bc7_profile("slow"); a.fastSkipTreshold_mode1 = 0; a.fastSkipTreshold_mode3 = 0
    e_full, m_full = errs(planar, full)
    e_a, _ = errs(planar, a)
    hist = np.bincount(m_full, minlength=8)
    print(f"{name:12s} blocks {nb}; winning modes {dict(enumerate(hist.tolist()))}; final error median {np.median(e_full):.0f}")
    for how in ("exact", 2, 1):
        lb13 = min_lb(tex, sub, range(64), 2, how, np.sqrt(3) / 2)
        lbs = 0.999 * lb13 - 1.0                      # the float-safety margin a kernel would carry
        need = lbs <= e_a
        wrong = ((m_full == 1) | (m_full == 3)) & ~need
        print(f"   LB {how}: phase B visits {100 * need.mean():5.1f} % of the blocks (blocks won by modes 1/3: {100 * ((m_full == 1) | (m_full == 3)).mean():5.1f} %; wrongly dropped {wrong.sum()})", flush=True)

if __name__ == "__main__":
    g = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(g, "inputs.npz")); z2 = np.load(os.path.join(g, "samples2.npz"))
    study("I3 smooth", surfaces.ldr_smooth(4096, 4096)[1024:1536, 2048:2560])
    for nm, a in (("baboon", z["baboon"]), ("monkey", z["monkey"][:216, :216]), ("colors260k", z2["colors260k"][:256, :256]),
                  ("normals", z2["normals"]), ("test_a", z2["test_a"][:256, :256]), ("landscape", z2["landscape_detail"][:336, :124])):
        study(nm, np.ascontiguousarray(a))
