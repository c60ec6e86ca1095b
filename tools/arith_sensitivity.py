"""Arithmetic-model sensitivity of the oracle.

The pinned model ("ISPC sse/avx target on an Intel CPU": LUT-seeded Newton rcp/rsqrt, K:1158 a true divide, no FMA)
rests on three statements about the ispc compiler that cannot be checked without an ispc binary.  This tool flips each
one in the ORACLE (test infrastructure; `make -C oracle variants`) and counts how many blocks change on the BASELINE
inputs, per format and preset:

    div1158rcp   `proj /= div` (kernel.ispc:1158) lowered as proj * rcp(div) like every other division
    ieee         rcp = 1.0f/v, rsqrt = 1.0f/sqrtf(v) (what an AMD host CPU or a portable build would be closest to)
    fma          gcc -ffp-contract=fast -mfma (the avx2 target's licence to fuse; gcc's choice of sums to fuse)
    ieee_fma     both
    reassoc      gcc -fassociative-math -freciprocal-math (-fno-signed-zeros -fno-trapping-math): sums and products may be
                 re-associated, reciprocals folded -- the other thing `--opt=fast-math` could license in LLVM

Output: a table of "% of blocks whose bytes differ from the pinned model" -> profiles/arith_sensitivity.txt.
Every variant is a legal encoding of the same quality class; the point is how far "bit-exact vs the ISPC binary" could
be off if an assumption is wrong, and therefore what the parity claim is worth on other hosts.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "intel-texture-works-plugin_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from oracle import pyoracle          # noqa: E402  (checker-side study)
from itw_amd import surfaces        # noqa: E402


def psnr(img, dec):
    d = img[..., :dec.shape[2]].astype(np.float64) - dec.astype(np.float64)
    return 10 * np.log10(255.0 ** 2 / max(np.mean(d * d), 1e-12))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=1024, help="edge of the synthetic crops (slow presets use half)")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "arith_sensitivity.txt"))
    a = ap.parse_args()
    n = a.size
    gold = dict(np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz")))
    ldr = {
        "I1 baboon.png 256^2": gold["baboon"],
        "I2 colors-16M crop": np.ascontiguousarray(surfaces.colors_16m(4096)[1024:1024 + n, 2048:2048 + n]),
        "I3 synthetic smooth+noise": surfaces.ldr_smooth(n, n),
        "I3u uniform random bytes": surfaces.ldr_uniform(n // 2, n // 2),
    }
    hdr = {
        "I4 monkey-32bit.hdr 220^2": gold["monkey_hdr"],
        "I4s synthetic HDR": surfaces.hdr_smooth(n // 2, n // 2),
        "I4r random half bits": surfaces.hdr_random_bits(n // 4, n // 4),
    }
    cases = [("bc1", None), ("bc3", None)] + [("bc7", p) for p in ("ultrafast", "veryfast", "basic", "slow", "alpha_basic", "alpha_slow")] \
        + [("bc6h", p) for p in ("veryfast", "fast", "basic", "slow")]
    slow = {"slow", "alpha_slow", "alpha_basic", "basic"}
    lines = []

    def emit(s=""):
        print(s, flush=True)
        lines.append(s)

    emit("# Arithmetic-model sensitivity of the oracle: % of blocks whose bytes differ from the pinned model")
    emit("# (tools/arith_sensitivity.py; variants = oracle/x86_math.h switches, `make -C oracle variants`)")
    emit(f"# host threads {pyoracle.usable_cores()}, crop edge {n} (half for basic/slow presets)")
    emit(f"{'format/preset':<20} {'input':<28} {'blocks':>8} " + " ".join(f"{v:>11}" for v in pyoracle.VARIANTS))
    t0 = time.time()
    totals = {}
    for fmt, prof in cases:
        for name, img in (hdr if fmt == "bc6h" else ldr).items():
            if prof in slow and img.shape[0] > n // 2:
                img = np.ascontiguousarray(img[:n // 2, :n // 2])
            h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
            img = np.ascontiguousarray(img[:h, :w])
            bpb = 8 if fmt == "bc1" else 16
            base = pyoracle.encode_mt(fmt, img, prof).reshape(-1, bpb)
            row = []
            for v in pyoracle.VARIANTS:
                with pyoracle.variant(v):
                    got = pyoracle.encode_mt(fmt, img, prof).reshape(-1, bpb)
                bad = int((got != base).any(axis=1).sum())
                row.append(100.0 * bad / base.shape[0])
                k = (fmt, prof, v)
                totals[k] = (totals.get(k, (0, 0))[0] + bad, totals.get(k, (0, 0))[1] + base.shape[0])
            emit(f"{fmt + ('/' + prof if prof else ''):<20} {name:<28} {base.shape[0]:>8} " + " ".join(f"{r:>10.3f}%" for r in row))
    emit()
    emit("# pooled over inputs")
    for fmt, prof in cases:
        emit(f"{fmt + ('/' + prof if prof else ''):<20} {'(all inputs)':<28} {totals[(fmt, prof, 'ieee')][1]:>8} "
             + " ".join(f"{100.0 * totals[(fmt, prof, v)][0] / totals[(fmt, prof, v)][1]:>10.3f}%" for v in pyoracle.VARIANTS))
    emit()
    emit("# quality is model-independent: PSNR (dB) of the decoded baboon.png stream under each model")
    img = gold["baboon"]
    for fmt, prof in (("bc1", None), ("bc7", "basic"), ("bc7", "slow")):
        vals = []
        for v in (None,) + pyoracle.VARIANTS:
            if v is None:
                blocks = pyoracle.encode(fmt, img, prof)
            else:
                with pyoracle.variant(v):
                    blocks = pyoracle.encode(fmt, img, prof)
            dec, _ = pyoracle.decode(fmt, blocks, img.shape[1], img.shape[0])
            vals.append(psnr(img[..., :3], dec[..., :3]))
        emit(f"{fmt + ('/' + prof if prof else ''):<20} pinned {vals[0]:.3f}  " + "  ".join(f"{n_} {x:.3f}" for n_, x in zip(pyoracle.VARIANTS, vals[1:])))
    emit(f"# wall {time.time() - t0:.0f} s")
    with open(a.out, "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
