"""The oracle's arithmetic-model switches (oracle/x86_math.h; study material for SURVEY 8c S2/S3, never parity targets).
The pinned model must stay the default; every variant must still produce legal streams, and on natural content they
must stay close to the pinned model (the committed study: profiles/arith_sensitivity.txt)."""
import struct

import numpy as np
import pytest


def _bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def test_default_library_is_the_pinned_lut_newton_model(oracle):
    L = oracle.lib()
    assert _bits(L.oracle_rcp(1.0)) == 0x3F7FFFFF          # NR on RCPPS(1) = 0x3f7ff000 is 1 ulp low
    assert _bits(L.oracle_rcp(16.0)) == 0x3D7FFFFF
    assert _bits(L.oracle_rsqrt(4.0)) != _bits(0.5)


def test_ieee_variant_computes_correctly_rounded_reciprocals(oracle):
    with oracle.variant("ieee"):
        L = oracle.lib()
        assert _bits(L.oracle_rcp(1.0)) == _bits(1.0) and _bits(L.oracle_rcp(3.0)) == _bits(np.float32(1.0) / np.float32(3.0))
        assert _bits(L.oracle_rsqrt(4.0)) == _bits(0.5)
    assert _bits(oracle.lib().oracle_rcp(1.0)) == 0x3F7FFFFF    # the context manager restored the pinned library


@pytest.mark.parametrize("variant", ["div1158rcp", "ieee", "fma", "ieee_fma"])
def test_variants_emit_legal_streams_close_to_the_pinned_model(oracle, golden_inputs, variant):
    img = golden_inputs["baboon"]
    hdr = golden_inputs["monkey_hdr"][:64, :64].copy()
    for fmt, prof, src, bpb, limit in (("bc1", None, img, 8, 0.03), ("bc7", "veryfast", img, 16, 0.03), ("bc6h", "fast", hdr, 16, 0.15)):
        base = oracle.encode(fmt, src, prof).reshape(-1, bpb)
        with oracle.variant(variant):
            got = oracle.encode(fmt, src, prof).reshape(-1, bpb)
        frac = float((got != base).any(axis=1).mean())
        assert frac <= limit, (fmt, prof, variant, frac)
        if variant == "div1158rcp" and fmt == "bc1":
            assert frac == 0.0                                   # kernel.ispc:1158 is not on the BC1 path
        dec, modes = oracle.decode(fmt, got.reshape(-1), src.shape[1], src.shape[0])
        assert (modes >= 0).all()
