"""FindClosestUNORM (BC4BC5.cpp:314-337) over its WHOLE domain: (red_0, red_1, texel code) in 256^3 = 16.7 M cases.

csrc/bc4_bc5.hip no longer searches 8 levels per texel: per endpoint pair the chosen index is piecewise constant in the texel
code, and the kernel counts run starts.  Here, on the CPU, exhaustively:
  1. the oracle's restatement of the search equals the REFERENCE's own function (static in BC4BC5.cpp, compiled unmodified into
     oracle/_ref/libdxtex_findclosest_ref.so) on every case;
  2. no endpoint pair has more than 8 runs (so 7 run starts + 8 run indices describe the function without loss), and the
     pairs red_0 != red_1 are monotone along the ramp;
  3. the kernel's evaluation of that table -- two texels per dword, carries out of byte additions -- reproduces every case.
The device-built table itself is compared with this one in tests/test_gpu_parity_bc4_bc5.py."""
import ctypes
import os

import numpy as np
import pytest

from _bc45_runs import evaluate, runs_table

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libdxtex_findclosest_ref.so")


@pytest.fixture(scope="module")
def F(oracle):
    return oracle.bc4_find_closest_table()


def test_oracle_search_equals_the_reference_function_on_all_16_7_million_cases(F):
    if not os.path.exists(REF):
        if os.path.exists("/root/reference/3rdParty/DirectXTex/DirectXTex/BC4BC5.cpp"):
            pytest.fail("oracle/_ref/libdxtex_findclosest_ref.so missing although /root/reference is here: run build()")
        pytest.skip("needs oracle/_ref (built from /root/reference)")
    L = ctypes.CDLL(REF)
    row = np.zeros(256, dtype=np.uint8)
    bad = 0
    for r0 in range(256):
        for r1 in range(256):
            L.dxtex_ref_find_closest_row(r0, r1, row.ctypes.data_as(ctypes.c_void_p))
            bad += int((row != F[r0, r1]).sum())
    assert bad == 0


def test_every_endpoint_pair_has_at_most_8_runs_and_the_table_form_is_lossless(F):
    table, max_runs = runs_table(F)
    assert max_runs == 8
    eq = np.arange(256) * 257                                                   # pairs red_0 == red_1: levels an ulp apart
    assert (table[eq, 1] >> 24).max() <= 5
    assert ((table[:, 1] >> 24) >= 1).all()
    flat = F.reshape(65536, 256)
    for got in evaluate(table):                                                 # each of the four texel positions of a step
        assert np.array_equal(got, flat)


def test_pairs_with_distinct_endpoints_are_monotone_along_the_ramp(F):
    """red_0 > red_1 (8 interpolated levels): ascending texel codes walk the indices 1,7,6,5,4,3,2,0; red_0 < red_1 (6 levels
    plus exact 0 and 1): 6,0,2,3,4,5,1,7 -- never back.  (Not needed by the table form; it is why 8 runs suffice.)"""
    pos8 = np.argsort(np.array([1, 7, 6, 5, 4, 3, 2, 0]))
    pos6 = np.argsort(np.array([6, 0, 2, 3, 4, 5, 1, 7]))
    r0, r1 = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    P = np.where((r0 > r1)[:, :, None], pos8[F], pos6[F]).astype(np.int8)
    mono = (np.diff(P, axis=2) >= 0).all(axis=2)
    assert mono[r0 != r1].all()
    assert (~mono).sum() == 70                                                  # all of them on the diagonal
