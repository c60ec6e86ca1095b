"""Synthetic input surfaces for parity tests and benchmarks (SURVEY.md section 8d inputs I3/I3u/I4s/I4r).

All generators are seeded and return numpy arrays laid out like an `rgba_surface`:
(H, W, 4) uint8 for BC1/BC3/BC7, (H, W, 4) uint16 (IEEE half bit patterns) for BC6H.
"""
import hashlib
import numpy as np

SEED = 20260925


def _smooth_fields(h, w, n_fields, rng, terms=3):
    """n_fields low-frequency 2-D cosine mixtures in roughly [-terms, terms]."""
    y = np.arange(h, dtype=np.float32)[:, None] / np.float32(h)
    x = np.arange(w, dtype=np.float32)[None, :] / np.float32(w)
    out = np.zeros((n_fields, h, w), dtype=np.float32)
    for f in range(n_fields):
        for _ in range(terms):
            fx, fy = rng.uniform(0.5, 24.0, size=2)
            ph = rng.uniform(0, 2 * np.pi)
            out[f] += np.cos(np.float32(2 * np.pi) * (np.float32(fx) * x + np.float32(fy) * y) + np.float32(ph))
    return out


def ldr_smooth(h, w, seed=SEED):
    """I3: clip(G + N, 0, 255); G = three low-frequency cosines per channel (amplitude 60 around 128),
    N ~ U{-24..24}; alpha is an independent field of the same kind (so alpha profiles see real alpha)."""
    rng = np.random.default_rng(seed)
    g = _smooth_fields(h, w, 4, rng) * np.float32(20.0) + np.float32(128.0)
    n = rng.integers(-24, 25, size=(4, h, w)).astype(np.float32)
    img = np.clip(np.rint(g + n), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(np.transpose(img, (1, 2, 0)))


def ldr_alpha_variant(img, kind, seed=SEED + 7):
    """The same RGB with another alpha channel, for the RGBA (`alpha_*`) BC7 profiles whose cost depends on it:
    "translucent" = as generated (every block has alpha well below 255), "opaque" = alpha 255 everywhere (what most of a
    real "has alpha" texture is), "mixed" = per 4x4 block either of the two, 50/50."""
    if kind == "translucent":
        return img
    out = img.copy()
    if kind == "opaque":
        out[..., 3] = 255
        return out
    if kind != "mixed":
        raise ValueError(kind)
    h, w = img.shape[:2]
    rng = np.random.default_rng(seed)
    opaque = rng.integers(0, 2, size=((h + 3) // 4, (w + 3) // 4), dtype=np.uint8).astype(bool)
    mask = np.repeat(np.repeat(opaque, 4, axis=0), 4, axis=1)[:h, :w]
    out[..., 3] = np.where(mask, np.uint8(255), img[..., 3])
    return out


def ldr_uniform(h, w, seed=SEED + 1):
    """I3u: i.i.d. uniform bytes -- every block spans the full range (worst case)."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)


def ldr_edge_cases(seed=SEED + 2):
    """A 64x64 (256 blocks) surface of hand-picked block classes: solid colours, two-colour blocks,
    1-D ramps, alpha extremes, single outlier texels, axis-aligned splits matching BC7 partitions."""
    rng = np.random.default_rng(seed)
    img = np.zeros((64, 64, 4), dtype=np.uint8)
    blocks = []
    for v in (0, 1, 127, 128, 254, 255):
        blocks.append(np.full((4, 4, 4), v, dtype=np.uint8))
    for _ in range(26):                                   # solid random colours, opaque and not
        c = rng.integers(0, 256, size=4, dtype=np.uint8)
        blocks.append(np.broadcast_to(c, (4, 4, 4)).copy())
    for _ in range(32):                                   # two colours, random split
        a, b = rng.integers(0, 256, size=(2, 4), dtype=np.uint8)
        m = rng.integers(0, 2, size=(4, 4, 1), dtype=np.uint8)
        blocks.append(np.where(m == 1, a, b).astype(np.uint8))
    for _ in range(32):                                   # linear ramps along x, y or the diagonal
        a, b = rng.integers(0, 256, size=(2, 4)).astype(np.float32)
        kind = rng.integers(0, 3)
        t = {0: np.arange(4)[None, :] / 3.0 + np.zeros((4, 1)),
             1: np.arange(4)[:, None] / 3.0 + np.zeros((1, 4)),
             2: (np.arange(4)[None, :] + np.arange(4)[:, None]) / 6.0}[int(kind)]
        blocks.append(np.clip(np.rint(a + (b - a) * t[..., None]), 0, 255).astype(np.uint8))
    for _ in range(32):                                   # alpha extremes over random colour
        blk = rng.integers(0, 256, size=(4, 4, 4), dtype=np.uint8)
        blk[..., 3] = rng.choice(np.array([0, 255], dtype=np.uint8), size=(4, 4))
        blocks.append(blk)
    for _ in range(32):                                   # smooth block + one outlier texel
        base = rng.integers(0, 256, size=4).astype(np.int32)
        blk = np.clip(base + rng.integers(-3, 4, size=(4, 4, 4)), 0, 255).astype(np.uint8)
        blk[rng.integers(0, 4), rng.integers(0, 4)] = rng.integers(0, 256, size=4, dtype=np.uint8)
        blocks.append(blk)
    for _ in range(32):                                   # left/right and top/bottom halves
        a, b = rng.integers(0, 256, size=(2, 4), dtype=np.uint8)
        blk = np.empty((4, 4, 4), dtype=np.uint8)
        if rng.integers(0, 2):
            blk[:, :2] = a; blk[:, 2:] = b
        else:
            blk[:2] = a; blk[2:] = b
        blk = np.clip(blk.astype(np.int32) + rng.integers(-2, 3, size=(4, 4, 4)), 0, 255).astype(np.uint8)
        blocks.append(blk)
    while len(blocks) < 256:                              # fully random filler
        blocks.append(rng.integers(0, 256, size=(4, 4, 4), dtype=np.uint8))
    for i, blk in enumerate(blocks[:256]):
        y, x = divmod(i, 16)
        img[y * 4:y * 4 + 4, x * 4:x * 4 + 4] = blk
    return img


def hdr_smooth(h, w, seed=SEED + 3):
    """I4s: smooth HDR radiance 2^field (about 2^-6 .. 2^6) with 5 % multiplicative noise, converted to
    half with round-to-nearest-even; alpha = 1.0 (0x3C00).  Mirrors what the plugin feeds BC6H (RGBA16F)."""
    rng = np.random.default_rng(seed)
    f = _smooth_fields(h, w, 3, rng) * np.float32(2.0)
    n = np.float32(1.0) + np.float32(0.05) * rng.standard_normal(size=(3, h, w)).astype(np.float32)
    rad = np.exp2(f) * np.abs(n)
    out = np.empty((h, w, 4), dtype=np.uint16)
    out[..., :3] = np.transpose(rad.astype(np.float16).view(np.uint16), (1, 2, 0))
    out[..., 3] = 0x3C00
    return out


def hdr_random_bits(h, w, seed=SEED + 4):
    """I4r: uniform random 16-bit patterns: negative halves, Inf, NaN and spans that overflow the int
    error accumulators (exercises the cvttps2dq / minps corner semantics).  Parity only, never benchmarked."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 65536, size=(h, w, 4), dtype=np.uint16)


COLORS_16M_SHA256 = "734d23cb367afaf0a40f4d4bcfc47088f0af7eb0d85109ee1c5809bd01dd9e85"   # sha256 of colors_16m(); equals the decoded PNG


def colors_16m(size=4096):
    """I2 of SURVEY.md 8d without the file: the reference's `Sample Images/colors-16M.png` (4096 x 4096, every 24-bit
    colour once) is R = x mod 256, G = y mod 256, B = 16 * (y div 256) + (x div 256); alpha = 255 as the plugin's
    RGB -> RGBA8 conversion fills it (IntelPlugin.cpp:741-766).  tests/test_oracle_bc1_bc3.py checks this formula
    against the PNG when the reference tree is present.  `size` < 4096 returns the top-left crop."""
    y, x = np.mgrid[0:size, 0:size]
    img = np.empty((size, size, 4), dtype=np.uint8)
    img[..., 0] = x % 256
    img[..., 1] = y % 256
    img[..., 2] = 16 * (y // 256) + x // 256
    img[..., 3] = 255
    return img


def tile_to(img, h, w):
    """Tile `img` up to (h, w) and crop."""
    ry = -(-h // img.shape[0])
    rx = -(-w // img.shape[1])
    return np.ascontiguousarray(np.tile(img, (ry, rx, 1))[:h, :w])


def sha256(arr):
    return hashlib.sha256(np.ascontiguousarray(arr).tobytes()).hexdigest()
