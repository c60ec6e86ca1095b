"""CPU: independent validity check of the oracle's bitstreams.  Every block the oracle emits must decode with a
from-spec decoder (oracle/bcn_decode.c, written from the format definition / the reference tree's own decoder
tables, not from the encoder) and reconstruct its source closely.  This is the one check the reference itself
applies to its output (the preview dialog decodes with DirectXTex)."""
import numpy as np
import pytest


def psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(255.0 ** 2 / max(mse, 1e-12))


@pytest.mark.parametrize("fmt,prof,floor", [("bc1", None, 29.0), ("bc3", None, 30.0), ("bc7", "ultrafast", 32.0),
                                            ("bc7", "basic", 36.0), ("bc7", "slow", 36.2), ("bc7", "alpha_basic", 37.0)])
def test_baboon_psnr(oracle, golden_inputs, golden_blocks, fmt, prof, floor):
    img = golden_inputs["baboon"]
    key = f"baboon.{fmt}" + (f".{prof}" if prof else "")
    blocks = golden_blocks[key] if key in golden_blocks else oracle.encode(fmt, img, prof)
    dec, modes = oracle.decode(fmt, blocks, 256, 256)
    assert (modes >= 0).all(), "a block used a reserved mode or did not consume exactly 128 bits"
    ch = 4 if (fmt == "bc3" or (prof or "").startswith("alpha")) else 3
    assert psnr(dec[..., :ch], img[..., :ch]) > floor


def test_bc7_all_modes_occur_and_decode(oracle, golden_inputs, golden_blocks):
    """The monkey photo (real alpha) under alpha_slow exercises modes 0-7; quality is monotone in the profile."""
    img = golden_inputs["monkey"]
    h, w = img.shape[:2]
    dec, modes = oracle.decode("bc7", golden_blocks["monkey.bc7.alpha_slow"], w, h)
    assert set(range(8)) <= set(modes.tolist())
    p_slow = psnr(dec, img)
    dec_f, m_f = oracle.decode("bc7", golden_blocks["monkey.bc7.alpha_ultrafast"], w, h)
    assert (m_f >= 0).all() and p_slow > psnr(dec_f, img) and p_slow > 45.0


def test_bc7_anchor_bits_are_implicit(oracle, golden_inputs, golden_blocks):
    """Re-encoding invariance: a decoder reading the anchor indices with one bit less must land on exactly
    128 bits for every block (mode return >= 0 asserts pos == 128 inside the decoder)."""
    for prof in ("basic", "slow", "alpha_slow"):
        _, modes = oracle.decode("bc7", golden_blocks[f"edge_cases.bc7.{prof}"], 64, 64)
        assert (modes >= 0).all()


@pytest.mark.parametrize("prof", ["veryfast", "fast", "basic", "slow", "veryslow"])
def test_bc6h_decodes_and_all_14_modes_occur(oracle, golden_inputs, golden_blocks, prof):
    img = golden_inputs["monkey_hdr"]
    h, w = img.shape[:2]
    dec, modes = oracle.decode("bc6h", golden_blocks[f"monkey_hdr.bc6h.{prof}"], w, h)
    assert (modes >= 0).all()
    if prof in ("slow", "veryslow"):
        assert set(range(14)) <= set(modes.tolist())
    f = lambda a: a.astype(np.uint16).view(np.float16).astype(np.float64)
    src = f(img[..., :3])
    rel = np.abs(f(dec) - src) / np.maximum(src, 1e-3)
    assert np.median(rel) < 0.01


def test_bc6h_random_bits_still_valid_blocks(oracle, golden_blocks):
    """Even for adversarial input every emitted block is a legal BC6H block."""
    for prof in ("fast", "slow"):
        _, modes = oracle.decode("bc6h", golden_blocks[f"hdr_random_bits.bc6h.{prof}"], 64, 32)
        assert (modes >= 0).all()
