"""CPU campaign (no GPU): the oracle's restatement (oracle/*.c) against the reference's own kernel.ispc compiled as a scalar
program (oracle/_ref/libispc_texcomp_ref_full.so, oracle/ref_build/ispc_as_cpp/), every format / preset on mixed content.
Usage: python tools/reference_kernel_campaign.py [megapixels_per_case]   -> one line per case + TOTAL; exit 1 on a mismatch."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np
from itw_amd import surfaces
from oracle import pyoracle, pyref

mp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
W = 1024
H = max(16, int(mp * 1e6 / W) // 16 * 16)
rng = np.random.default_rng(2027)
q = H // 4 // 4 * 4


def posterised(h, w, levels, seed):
    img = surfaces.ldr_smooth(h, w, seed=surfaces.SEED + seed)
    return (img // (256 // levels) * (256 // levels) + 256 // levels // 2).astype(np.uint8)


smooth = surfaces.ldr_smooth(q, W, seed=surfaces.SEED + 51).copy()
smooth[:, W // 4:W // 2, 3] = 255
smooth[:, W // 2:3 * W // 4, 3] = np.where(rng.random((q, W // 4)) < 0.06, 254, 255).astype(np.uint8)
ldr = np.ascontiguousarray(np.concatenate([smooth, rng.integers(0, 256, (q, W, 4), dtype=np.uint8), posterised(q, W, 4, 52),
                                           surfaces.colors_16m()[2048:2048 + H - 3 * q, 1024:1024 + W]], axis=0))
hq = H // 2 // 4 * 4
hdr = np.ascontiguousarray(np.concatenate([surfaces.hdr_smooth(hq, W, seed=surfaces.SEED + 53),
                                           rng.integers(0, 65536, (H - hq, W, 4), dtype=np.uint16)], axis=0))
cases = [("bc1", None), ("bc3", None)] + [("bc7", p) for p in ("ultrafast", "veryfast", "fast", "basic", "slow", "alpha_ultrafast",
         "alpha_veryfast", "alpha_fast", "alpha_basic", "alpha_slow")] + [("bc6h", p) for p in ("veryfast", "fast", "basic", "slow", "veryslow")]
bad_total = blocks_total = 0
for fmt, prof in cases:
    img = hdr if fmt == "bc6h" else ldr
    t0 = time.perf_counter()
    a = pyref.encode_mt(fmt, img, prof)
    t1 = time.perf_counter()
    b = pyoracle.encode_mt(fmt, img, prof).reshape(-1)
    t2 = time.perf_counter()
    bpb = 8 if fmt == "bc1" else 16
    bad = int((a.reshape(-1, bpb) != b.reshape(-1, bpb)).any(axis=1).sum())
    n = a.size // bpb
    bad_total += bad; blocks_total += n
    print(f"{fmt:5s} {prof or '-':16s} {n:8d} blocks  mismatches {bad:6d}   kernel.ispc (scalar) {t1 - t0:6.1f} s  oracle {t2 - t1:6.1f} s", flush=True)
print(f"TOTAL {blocks_total} blocks, {bad_total} mismatches  (content: smooth + opaque / speckled alpha, random bytes, posterised, colors-16M; HDR smooth + random half bits)")
sys.exit(1 if bad_total else 0)
