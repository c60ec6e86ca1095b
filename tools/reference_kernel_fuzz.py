"""CPU fuzz (no GPU): random settings structs x adversarial block classes through the oracle's restatement and through the
reference's own kernel.ispc (scalar build, oracle/_ref/libispc_texcomp_ref_full.so).  Usage:
python tools/reference_kernel_fuzz.py [trials] [seed]  -> summary line; exit 1 on the first mismatch (prints the settings)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np
from itw_amd import surfaces
from oracle import pyoracle, pyref

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)


def ldr_classes(n):
    """n blocks per class, one block row each: flat, two colours, gradients, extremes, noise on a ramp, alpha cut-outs."""
    rows = []
    w = 4 * n
    flat = np.repeat(np.repeat(rng.integers(0, 256, (1, n, 4), dtype=np.uint8), 4, 0), 4, 1)
    two = np.where(rng.random((4, w, 1)) < 0.5, np.repeat(rng.integers(0, 256, (1, n, 4)), 4, 1), np.repeat(rng.integers(0, 256, (1, n, 4)), 4, 1)).astype(np.uint8)
    x = np.linspace(0, 255, w)[None, :, None] * np.ones((4, 1, 4))
    grad = (x * rng.random((1, 1, 4)) + np.arange(4)[:, None, None] * 9).clip(0, 255).astype(np.uint8)
    ext = rng.choice(np.array([0, 1, 127, 128, 254, 255], dtype=np.uint8), (4, w, 4))
    noise = (x + rng.integers(-6, 7, (4, w, 4))).clip(0, 255).astype(np.uint8)
    cut = surfaces.ldr_smooth(4, w, seed=int(rng.integers(1, 1 << 20))).copy()
    cut[..., 3] = np.where(rng.random((4, w)) < 0.5, 0, 255)
    near = surfaces.ldr_smooth(4, w, seed=int(rng.integers(1, 1 << 20))).copy()
    near[..., 3] = rng.integers(250, 256, (4, w))
    for r in (flat, two, grad, ext, noise, cut, near, rng.integers(0, 256, (4, w, 4), dtype=np.uint8)):
        rows.append(r)
    return np.ascontiguousarray(np.concatenate(rows, axis=0))


def hdr_classes(n):
    w = 4 * n
    smooth = surfaces.hdr_smooth(4, w, seed=int(rng.integers(1, 1 << 20))).view(np.uint16)
    bits = rng.integers(0, 65536, (4, w, 4), dtype=np.uint16)
    flat = np.repeat(np.repeat(rng.integers(0, 0x7c00, (1, n, 4), dtype=np.uint16), 4, 0), 4, 1)
    small = rng.integers(0, 64, (4, w, 4), dtype=np.uint16)
    big = rng.integers(0x7800, 0x7c00, (4, w, 4), dtype=np.uint16)
    narrow = (np.uint16(0x3c00) + rng.integers(0, 40, (4, w, 4))).astype(np.uint16)
    return np.ascontiguousarray(np.concatenate([smooth, bits, flat, small, big, narrow], axis=0))


thr = [0, 1, 2, 3, 5, 12, 16, 17, 33, 63, 64]
blocks = 0
for t in range(trials):
    img = ldr_classes(16)
    s = pyoracle.Bc7Settings()
    s.skip_mode2 = bool(rng.integers(0, 2))
    s.fastSkipTreshold_mode1, s.fastSkipTreshold_mode3, s.fastSkipTreshold_mode7 = (int(rng.choice(thr)) for _ in range(3))
    s.mode45_channel0 = int(rng.integers(0, 4)); s.refineIterations_channel = int(rng.integers(0, 6)); s.channels = int(rng.choice([3, 4]))
    sel = [bool(rng.integers(0, 2)) for _ in range(4)]
    if not any(sel): sel[int(rng.integers(0, 4))] = True
    for i in range(4): s.mode_selection[i] = sel[i]
    for i in range(8): s.refineIterations[i] = int(rng.integers(0, 6))
    a, b = pyref.encode("bc7", img, s), pyoracle.encode("bc7", img, s).reshape(-1)
    if not np.array_equal(a, b):
        bad = np.nonzero((a.reshape(-1, 16) != b.reshape(-1, 16)).any(axis=1))[0]
        print("BC7 MISMATCH trial", t, "blocks", bad[:8], bytes(s).hex()); sys.exit(1)
    for fmt in ("bc1", "bc3"):
        if not np.array_equal(pyref.encode(fmt, img), pyoracle.encode(fmt, img).reshape(-1)):
            print(fmt, "MISMATCH trial", t); sys.exit(1)
    h = hdr_classes(16)
    s6 = pyoracle.Bc6hSettings()
    s6.slow_mode, s6.fast_mode = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    s6.refineIterations_1p, s6.refineIterations_2p = int(rng.integers(0, 4)), int(rng.integers(0, 4))
    s6.fastSkipTreshold = int(rng.choice([0, 1, 2, 4, 10, 31, 32]))
    a, b = pyref.encode("bc6h", h, s6), pyoracle.encode("bc6h", h, s6).reshape(-1)
    if not np.array_equal(a, b):
        bad = np.nonzero((a.reshape(-1, 16) != b.reshape(-1, 16)).any(axis=1))[0]
        print("BC6H MISMATCH trial", t, "blocks", bad[:8], bytes(s6).hex()); sys.exit(1)
    blocks += 3 * (img.shape[0] // 4) * 16 + (h.shape[0] // 4) * 16
print(f"{trials} random bc7_enc_settings + {trials} random bc6h_enc_settings structs x 8 LDR / 6 HDR block classes (flat, two-colour, gradient, extreme codes, "
      f"ramp + noise, alpha cut-outs, near-opaque alpha, random; HDR smooth, random bits, flat, denormal-small, near-max, narrow): {blocks} blocks, 0 mismatches")
