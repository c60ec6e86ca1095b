"""Builds tests/golden/*.npz.  Run HERE (the build container): it reads the reference's sample images
from /root/reference/Sample Images (inputs only -- the reference ships no expected outputs), converts them to
the surface layouts the C ABI takes, and records the ORACLE's output for each (format, profile).

The golden outputs are therefore oracle-generated, not reference-generated: they pin the oracle (and through
the parity tests the HIP kernels) against regressions; they do not prove equality with an ISPC binary
(none can be built here: no ispc compiler).  That limit is stated in DESIGN.md ("parity unpinned").

    python tools/make_golden.py            # writes inputs + golden block streams
"""
import os
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from oracle import pyoracle  # noqa: E402
from itw_amd import surfaces  # noqa: E402
import rgbe  # noqa: E402

SAMPLES = "/root/reference/Sample Images"
OUT = os.path.join(ROOT, "tests", "golden")


def load_png_rgba(name):
    from PIL import Image
    im = Image.open(os.path.join(SAMPLES, name)).convert("RGBA")
    return np.ascontiguousarray(np.asarray(im, dtype=np.uint8))


def float_to_half_bits(rgb):
    """float32 -> half, round-to-nearest-even, like XMConvertFloatToHalf (the plugin's conversion,
    IntelPlugin.cpp:343-366); alpha = 1.0."""
    h = rgb.astype(np.float16).view(np.uint16)
    out = np.empty(rgb.shape[:2] + (4,), dtype=np.uint16)
    out[..., :3] = h
    out[..., 3] = 0x3C00
    return out


def pad4(img):
    """Edge replication to a multiple of 4 (the plugin's padding rule, IntelPlugin.cpp:893-928)."""
    h, w = img.shape[:2]
    return np.pad(img, ((0, (-h) % 4), (0, (-w) % 4), (0, 0)), mode="edge")


def main():
    os.makedirs(OUT, exist_ok=True)
    pyoracle.build()
    inputs = {
        "baboon": load_png_rgba("baboon.png"),                       # 256x256, alpha == 255  (BASELINE configs[0])
        "monkey": pad4(load_png_rgba("monkey.png")),                 # 220x220, real alpha
        "edge_cases": surfaces.ldr_edge_cases(),                     # 64x64 hand-picked block classes
    }
    hdr = {
        "monkey_hdr": pad4(float_to_half_bits(rgbe.read_hdr(os.path.join(SAMPLES, "monkey-32bit.hdr")))),
        "hdr_random_bits": surfaces.hdr_random_bits(32, 64),
    }
    np.savez_compressed(os.path.join(OUT, "inputs.npz"), **inputs, **hdr)
    gold = {}
    for name, img in inputs.items():
        for fmt in ("bc1", "bc3"):
            gold[f"{name}.{fmt}"] = pyoracle.encode(fmt, img)
        if pyoracle.has("oracle_CompressBlocksBC7"):
            profs = ["ultrafast", "veryfast", "fast", "basic", "slow",
                     "alpha_ultrafast", "alpha_veryfast", "alpha_fast", "alpha_basic", "alpha_slow"]
            for p in profs:
                if name == "baboon" and p not in ("veryfast", "basic", "slow", "alpha_basic"):
                    continue
                gold[f"{name}.bc7.{p}"] = pyoracle.encode("bc7", img, p)
    if pyoracle.has("oracle_CompressBlocksBC6H"):
        for name, img in hdr.items():
            for p in ("veryfast", "fast", "basic", "slow", "veryslow"):
                gold[f"{name}.bc6h.{p}"] = pyoracle.encode("bc6h", img, p)
    np.savez_compressed(os.path.join(OUT, "golden_blocks.npz"), **gold)
    for k, v in sorted(gold.items()):
        print(f"{k:40s} {v.size:8d} B  sha256 {surfaces.sha256(v)[:16]}")


def main_bc45():
    """BC4/BC5 (the DirectXTex formats): separate file so the ISPC-format fixtures stay byte-identical.  Includes the
    217 x 215 and 5 x 7 crops of monkey.png (odd sizes: DirectXTex's partial-block rule on both edges)."""
    pyoracle.build()
    monkey = load_png_rgba("monkey.png")
    data = {"monkey_crop.input": np.ascontiguousarray(monkey[1:218, 2:217]),
            "tiny.input": np.ascontiguousarray(monkey[40:45, 60:67])}
    cases = {"baboon": load_png_rgba("baboon.png"), "edge_cases": surfaces.ldr_edge_cases(),
             "monkey_crop": data["monkey_crop.input"], "tiny": data["tiny.input"]}
    for name, img in cases.items():
        for fmt in ("bc4", "bc5"):
            data[f"{name}.{fmt}"] = pyoracle.encode_bc45(fmt, img)
    np.savez_compressed(os.path.join(OUT, "golden_bc45.npz"), **data)
    for k, v in sorted(data.items()):
        print(f"{k:40s} {v.size:8d} B  sha256 {surfaces.sha256(v)[:16]}")


if __name__ == "__main__":
    if "--bc45" in sys.argv:
        main_bc45()
    else:
        main()
