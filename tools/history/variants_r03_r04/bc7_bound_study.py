"""Feasibility study (CPU, oracle as the source of the incumbent): how many (block, shape) pairs of the BC7 scans could an EXACT
lower bound of part_fast's error prune?  Bound: part_fast's palette is a set of rounded points on a segment, so a subset's error is
>= (sqrt(R) - delta sqrt(n))_+^2 with R the PCA residual of the subset (sum of squared distances to the best line), delta = sqrt(3)/2.
Optimistic incumbent = the block's FINAL error (decoded oracle output).  Prints survivors per block and per 64-block wave."""
import os, sys, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
from oracle import pyoracle
from itw_amd import surfaces

def tables():
    t = open(os.path.join(ROOT, "oracle", "bc7_tables.h")).read()
    def arr(name):
        s = t.index(name); s = t.index("{", s); e = t.index("}", s)
        return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\d+", re.sub(r"//[^\n]*", "", t[s:e]))]
    names = re.findall(r"(\w+)\s*\[", t)
    return t, names

def subsets():
    """[128][16] subset id per texel: 0-63 two-subset shapes, 64-127 three-subset shapes (from the pattern words)."""
    t = open(os.path.join(ROOT, "oracle", "bc7_tables.h")).read()
    m = re.search(r"BCN_PATTERN\[128\]\s*=\s*\{([^}]*)\}", t)
    vals = [int(x, 16) for x in re.findall(r"0x([0-9a-fA-F]+)u", m.group(1))]
    assert len(vals) == 128, len(vals)
    return np.array([[(v >> (2 * k)) & 3 for k in range(16)] for v in vals])

def study(name, img):
    h, w = img.shape[:2]
    blocks = pyoracle.encode("bc7", img, "slow")
    dec = pyoracle.decode("bc7", blocks, w, h)[0]
    t = img[..., :3].astype(np.float64); d = dec[..., :3].astype(np.float64)
    e = ((t - d) ** 2).reshape(h // 4, 4, w // 4, 4, 3).sum(axis=(1, 3, 4))          # [by, bx] final error
    tex = t.reshape(h // 4, 4, w // 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 3)   # [block, texel, c]
    best = e.reshape(-1)
    sub = subsets()
    nb = tex.shape[0]
    lb = np.zeros((nb, 128))
    for p in range(128):
        for s in range(3 if p >= 64 else 2):
            m = sub[p] == s
            n = m.sum()
            x = tex[:, m, :]
            x = x - x.mean(axis=1, keepdims=True)
            c = np.einsum("bki,bkj->bij", x, x)
            lam = np.linalg.eigvalsh(c)[:, -1]
            r = np.maximum(np.trace(c, axis1=1, axis2=2) - lam, 0)
            lb[:, p] += np.maximum(np.sqrt(r) - np.sqrt(3) / 2 * np.sqrt(n), 0) ** 2
    for label, sl in (("modes 1/3 (64 two-subset shapes)", slice(0, 64)), ("mode 2 (64 three-subset shapes)", slice(64, 128)), ("mode 0 (16 three-subset shapes)", slice(64, 80))):
        surv = lb[:, sl] < best[:, None] + 1e-9
        per_block = surv.sum(axis=1)
        nw = nb // 64
        per_wave_max = per_block[:nw * 64].reshape(nw, 64).max(axis=1)
        anyw = surv[:nw * 64].reshape(nw, 64, -1).any(axis=1).sum(axis=1)
        n = surv.shape[1]
        print(f"{name:14s} {label:34s} survivors/block mean {per_block.mean():5.1f} of {n}  ({100 * per_block.mean() / n:4.1f} %)   "
              f"wave max-lane mean {per_wave_max.mean():5.1f} ({100 * per_wave_max.mean() / n:4.1f} %)   shapes some lane of the wave needs {anyw.mean():5.1f}", flush=True)
    print(f"{name:14s} final error per block: median {np.median(best):.0f}, mean {best.mean():.0f}")

if __name__ == "__main__":
    img = surfaces.ldr_smooth(1024, 1024)
    study("I3 ldr_smooth", img)
    g = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(g, "inputs.npz")); z2 = np.load(os.path.join(g, "samples2.npz"))
    for nm, a in (("baboon", z["baboon"]), ("monkey", z["monkey"][:216, :216]), ("colors260k", z2["colors260k"][:256, :256]),
                  ("normals", z2["normals"]), ("test_a", z2["test_a"][:256, :256])):
        study(nm, np.ascontiguousarray(a))
