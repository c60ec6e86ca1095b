"""The pinned arithmetic of the oracle (oracle/x86_math.h): Newton-refined rcp/rsqrt on the Intel seed tables,
cvttps2dq truncation.  Known answers are the values SURVEY.md Appendix B measured on an Intel Xeon with the real
instructions; on an Intel host the extraction tool re-proves the table model against RCPPS/RSQRTPS for all 2^32 inputs."""
import os
import struct
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(f):
    return struct.unpack("<I", struct.pack("<f", f))[0]


def fl(u):
    return struct.unpack("<f", struct.pack("<I", u))[0]


def test_seed_known_answers(oracle):
    L = oracle.lib()
    assert bits(L.oracle_rcpps(1.0)) == 0x3F7FF000
    assert bits(L.oracle_rcpps(3.0)) == 0x3EAAA000
    assert bits(L.oracle_rsqrtps(1.0)) == 0x3F7FF000
    assert bits(L.oracle_rsqrtps(2.0)) == 0x3F34F800


def test_newton_step_known_answers(oracle):
    L = oracle.lib()
    assert bits(L.oracle_rcp(1.0)) == 0x3F7FFFFF            # rcp(1) != 1
    assert bits(L.oracle_rcp(16.0)) == 0x3D7FFFFF
    for n in range(1, 17):                                  # always one ulp below 1/n (Appendix B)
        assert bits(L.oracle_rcp(float(n))) == bits(np.float32(1.0) / np.float32(n)) - 1, n
    assert np.isnan(L.oracle_rcp(0.0))                      # inf * (2 - 0*inf) = NaN
    assert np.isnan(L.oracle_rcp(float("inf")))
    assert np.isnan(L.oracle_rsqrt(0.0))
    assert np.isnan(L.oracle_rsqrt(-1.0))


def test_seed_specials(oracle):
    L = oracle.lib()
    assert L.oracle_rcpps(0.0) == float("inf") and L.oracle_rcpps(-0.0) == float("-inf")
    assert L.oracle_rcpps(fl(0x00000001)) == float("inf")           # denormal operand treated as zero
    assert bits(L.oracle_rcpps(float("inf"))) == 0 and bits(L.oracle_rcpps(float("-inf"))) == 0x80000000
    assert bits(L.oracle_rcpps(fl(0x7F000000))) == 0                # 2^127 -> result denormal -> flushed
    assert bits(L.oracle_rsqrtps(float("inf"))) == 0
    assert bits(L.oracle_rsqrtps(-4.0)) == 0xFFC00000
    assert L.oracle_rsqrtps(-0.0) == float("-inf")


def test_newton_accuracy(oracle):
    """One NR step brings the 12-bit seeds to ~23 bits: relative error < 2^-22 on a mantissa sweep."""
    L = oracle.lib()
    xs = np.float32(1.0) + np.arange(0, 1 << 23, 4099, dtype=np.float64).astype(np.float32) / np.float32(1 << 23)
    for scale in (np.float32(1.0), np.float32(1e-3), np.float32(7e4)):
        for x in (xs[::37] * scale):
            r = L.oracle_rcp(float(x))
            assert abs(r * float(x) - 1.0) < 2.0 ** -21
            s = L.oracle_rsqrt(float(x))
            assert abs(s * s * float(x) - 1.0) < 2.0 ** -20


def test_f2i_is_cvttps2dq(oracle):
    L = oracle.lib()
    INT_MIN = -2147483648
    assert L.oracle_f2i(1.9) == 1 and L.oracle_f2i(-1.9) == -1
    assert L.oracle_f2i(2147483520.0) == 2147483520
    for bad in (float("nan"), float("inf"), float("-inf"), 2147483648.0, 3e9, -3e9):
        assert L.oracle_f2i(bad) == INT_MIN
    assert L.oracle_f2i(-2147483648.0) == INT_MIN


def test_table_model_matches_the_cpu_instructions_on_intel_hosts(tmp_path):
    """Exhaustive (2^32 inputs) check of the LUT model against RCPPS/RSQRTPS, and the committed tables are what
    the tool regenerates.  Only meaningful on Intel CPUs (AMD seeds differ)."""
    cpuinfo = open("/proc/cpuinfo").read() if os.path.exists("/proc/cpuinfo") else ""
    if "GenuineIntel" not in cpuinfo:
        pytest.skip("host CPU is not Intel: hardware seeds differ from the committed Intel tables")
    exe = tmp_path / "xlut"
    subprocess.run(["gcc", "-O2", "-msse2", "-fopenmp", os.path.join(ROOT, "tools", "extract_x86_luts.c"), "-o", str(exe)],
                   check=True)
    a, b = tmp_path / "a.h", tmp_path / "b.h"
    r = subprocess.run([str(exe), str(a), str(b)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "rcp mismatches 0, rsqrt mismatches 0" in r.stderr
    assert a.read_text() == open(os.path.join(ROOT, "oracle", "x86_luts.h")).read()
    assert b.read_text() == open(os.path.join(ROOT, "intel-texture-works-plugin_amd", "csrc", "x86_luts_packed.h")).read()
