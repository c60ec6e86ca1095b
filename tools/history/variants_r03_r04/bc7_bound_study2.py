"""Feasibility study 2 (CPU): wave-level branch-and-bound inside ONE mode's shape scan.
A shape p can be skipped for a wave when every lane's lower bound LB(p) is above that lane's best part_fast error so far (strict >,
so no tie-break is touched).  LB: a subset's palette is rounded points on a segment => err_s >= (sqrt(R_s) - delta sqrt(n_s))_+^2, R_s = PCA
residual.  Simulates the scan of 64-block waves (raster order) with uniform shape order, using the oracle's own part_fast errors."""
import os, sys, re, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
from oracle import pyoracle
from itw_amd import surfaces

def subsets():
    t = open(os.path.join(ROOT, "oracle", "bc7_tables.h")).read()
    m = re.search(r"BCN_PATTERN\[128\]\s*=\s*\{([^}]*)\}", t)
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)u", m.group(1))]
    return np.array([[(v >> (2 * k)) & 3 for k in range(16)] for v in vals])

def lam_upper(c, squarings):
    """upper bound of the largest eigenvalue of PSD c: ||c^(2^s)||_F ^ (1 / 2^s)"""
    m = c
    for _ in range(squarings):
        m = m @ m
    return np.sqrt((m * m).sum(axis=(1, 2))) ** (1.0 / (1 << squarings))

def bounds(tex, sub, shapes, nsub, how):
    nb = tex.shape[0]
    lb = np.zeros((nb, len(shapes)))
    for i, p in enumerate(shapes):
        for s in range(nsub):
            m = sub[p] == s
            n = m.sum()
            x = tex[:, m, :]
            x = x - x.mean(axis=1, keepdims=True)
            c = np.einsum("bki,bkj->bij", x, x)
            tr = np.trace(c, axis1=1, axis2=2)
            lam = np.linalg.eigvalsh(c)[:, -1] if how == "exact" else lam_upper(c, how)
            r = np.maximum(tr - lam, 0)
            lb[:, i] += np.maximum(np.sqrt(r) - np.sqrt(3) / 2 * np.sqrt(n), 0) ** 2
    return lb

def study(name, img):
    L = pyoracle.lib()
    fn = L.oracle_bc7_part_fast_errors
    fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]; fn.restype = None
    h, w = img.shape[:2]
    tex = img[..., :3].astype(np.float64).reshape(h // 4, 4, w // 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 3)
    nb = tex.shape[0]
    planar = np.zeros((nb, 64), dtype=np.float32)
    planar[:, :48] = tex.transpose(0, 2, 1).reshape(nb, 48)
    planar[:, 48:] = 255
    sub = subsets()
    err = np.zeros(64, dtype=np.float32); key = np.zeros(64, dtype=np.int32)
    for mode in (3, 1, 2, 0):
        nsub = 3 if mode in (0, 2) else 2
        count = 16 if mode == 0 else 64
        E = np.zeros((nb, count)); K = np.zeros((nb, count), dtype=np.int64)
        for b in range(nb):
            fn(planar[b].ctypes.data, mode, err.ctypes.data, key.ctypes.data)
            E[b] = err[:count]; K[b] = key[:count]
        shapes = [p + (64 if nsub == 3 else 0) for p in range(count)]
        out = []
        for how in ("exact", 2, 1):
            lb = bounds(tex, sub, shapes, nsub, how)
            assert (lb <= E + 1e-6).all(), (name, mode, how, float((lb - E).max()))
            nw = nb // 64
            lbw = lb[:nw * 64].reshape(nw, 64, count); Ew = E[:nw * 64].reshape(nw, 64, count)
            for order_name in ("index", "wave-sum-of-LB"):
                evaluated = 0
                lane_needed = 0
                for wv in range(nw):
                    order = np.arange(count) if order_name == "index" else np.argsort(lbw[wv].sum(axis=0), kind="stable")
                    inc = np.full(64, np.inf)
                    for p in order:
                        need = lbw[wv, :, p] <= inc
                        lane_needed += need.sum()
                        if need.any():
                            evaluated += 1
                            inc = np.minimum(inc, Ew[wv, :, p])
                out.append(f"{how}/{order_name}: wave evaluates {100 * evaluated / (nw * count):5.1f} % (lanes needing {100 * lane_needed / (nw * 64 * count):5.1f} %)")
        print(f"{name:12s} mode {mode}: " + " | ".join(out), flush=True)

if __name__ == "__main__":
    g = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(g, "inputs.npz")); z2 = np.load(os.path.join(g, "samples2.npz"))
    study("I3 smooth", surfaces.ldr_smooth(512, 512))
    for nm, a in (("baboon", z["baboon"]), ("monkey", z["monkey"][:216, :216]), ("colors260k", z2["colors260k"][:256, :256]),
                  ("normals", z2["normals"]), ("test_a", z2["test_a"][:256, :256])):
        study(nm, np.ascontiguousarray(a))
