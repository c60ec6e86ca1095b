"""CPU feasibility study: how often can modes 4/5/6 be ruled out EXACTLY after modes 0/2 under an RGB profile?

A mode-6 encoding is rounded points of ONE segment for the whole block (kernel.ispc:1651-1684): error >= (sqrt(R3) - sqrt(3)/2 * 4)_+^2 with R3
the residual of the 16 texels about their best line in RGB.  A mode-4/5 candidate of rotation p (channels == 3: p = channel0..2,
kernel.ispc:1572-1586) replaces channel p by the constant 255 in the vector part and codes channel p as the scalar part: the vector part is
rounded points of one segment in the two remaining channels, error >= (sqrt(R2(p)) - sqrt(2)/2 * 4)_+^2, and the scalar part's error is >= 0
(optionally: >= the best 1-D quantisation bound below).  Modes 4,5,6 only replace the block on a strict `<`, so they can be skipped where
that bound is >= the error modes 0/2 left.  Incumbent = the oracle's modes-{0,2}-only encode (exact error from its decoded blocks).
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
from oracle import pyoracle
from itw_amd import surfaces


def block_err(img, blocks):
    h, w = img.shape[:2]
    dec = pyoracle.decode("bc7", blocks, w, h)[0]
    t = img[..., :3].astype(np.float64); d = dec[..., :3].astype(np.float64)
    return ((t - d) ** 2).reshape(h // 4, 4, w // 4, 4, 3).sum(axis=(1, 3, 4)).reshape(-1)


def settings(sel):
    s = pyoracle.bc7_profile("slow")
    for i in range(4):
        s.mode_selection[i] = sel[i]
    return s


def line_residual(x):
    x = x - x.mean(axis=1, keepdims=True)
    c = np.einsum("bki,bkj->bij", x, x)
    lam = np.linalg.eigvalsh(c)[:, -1]
    return np.maximum(np.trace(c, axis1=1, axis2=2) - lam, 0)


def scalar_bound(v, levels):
    """cheap exact lower bound of a 1-D channel coded with `levels` equally spaced interpolants between two byte endpoints, each decoded
    value rounded to an integer: the decoded values lie within 1/2 of `levels` collinear, equally spaced reals; lower bound used here = 0
    unless the channel has more distinct values than levels (then at least ... kept 0: the study reports the zero-scalar bound)."""
    return np.zeros(v.shape[0])


def study(name, img, out):
    h, w = img.shape[:2]
    e02 = block_err(img, pyoracle.encode("bc7", img, settings((1, 0, 0, 0))))
    e0246 = block_err(img, pyoracle.encode("bc7", img, settings((1, 0, 1, 1))))
    efull = block_err(img, pyoracle.encode("bc7", img, "slow"))
    tex = img[..., :3].astype(np.float64).reshape(h // 4, 4, w // 4, 4, 3).transpose(0, 2, 1, 3, 4).reshape(-1, 16, 3)
    b6 = np.maximum(np.sqrt(line_residual(tex)) - np.sqrt(3) / 2 * 4, 0) ** 2
    b45 = np.full(tex.shape[0], np.inf)
    for p in range(3):
        keep = [c for c in range(3) if c != p]
        b = np.maximum(np.sqrt(line_residual(tex[:, :, keep])) - np.sqrt(2) / 2 * 4, 0) ** 2
        b45 = np.minimum(b45, b)
    skip6 = b6 >= e02
    skip45 = b45 >= e02
    won456 = e0246 < e02
    nb = len(e02)
    nw = nb // 64
    both = skip6 & skip45
    wave_all = both[:nw * 64].reshape(nw, 64).all(axis=1).mean()
    line = (f"{name:14s} blocks {nb:6d} | modes 4/5/6 actually improve on 0/2: {100 * won456.mean():5.1f} % | skip mode 6: {100 * skip6.mean():5.1f} %  "
            f"skip modes 4/5: {100 * skip45.mean():5.1f} %  skip all of 4/5/6: {100 * both.mean():5.1f} %  (whole waves of 64: {100 * wave_all:5.1f} %) | "
            f"median e02 {np.median(e02):.0f}  median bound6 {np.median(b6):.0f}  median bound45 {np.median(b45):.0f}  median final {np.median(efull):.0f}")
    assert not (both & won456).any(), "bound violated"
    print(line, flush=True)
    out.append(line)


if __name__ == "__main__":
    out = []
    g = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(g, "inputs.npz")); z2 = np.load(os.path.join(g, "samples2.npz"))
    study("I3 ldr_smooth", surfaces.ldr_smooth(512, 512), out)
    for nm, a in (("baboon", z["baboon"][:256, :256]), ("monkey", z["monkey"][:216, :216]), ("colors260k", z2["colors260k"][:256, :256]),
                  ("normals", z2["normals"]), ("test_a", z2["test_a"][:256, :256])):
        study(nm, np.ascontiguousarray(a), out)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(__doc__ + "\n" + "\n".join(out) + "\n")
