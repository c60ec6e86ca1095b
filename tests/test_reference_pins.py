"""What the reference itself pins.  Everything of the path that is plain C++ is compiled UNMODIFIED from /root/reference
into oracle/_ref/ (recipe: oracle/ref_build/Makefile; the binaries travel to the GPU box, the sources never enter the
repo):

  * ispc_texcomp.cpp:20-440  -> the 15 presets + ABI wrappers (libispc_texcomp_ref.so; kernel entry points = oracle)
  * ispc_texcomp.h:19-107    -> a caller compiled against the reference's header, linked against the PRODUCT library
  * win32Threads.cpp:192-329 -> the reference's own CompressImageMT/ST + 17 trampolines (through a pthread Win32 shim)
                                driving the product library with host pointers: "the plugin calls the ABI unchanged"

kernel.ispc itself is covered by tests/test_reference_kernel_source.py (built as one scalar program instance).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import first_mismatch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
REFERENCE_TREE = "/root/reference/3rdParty/Intel/Source/ispc_texcomp.cpp"
BC7 = ["ultrafast", "veryfast", "fast", "basic", "slow",
       "alpha_ultrafast", "alpha_veryfast", "alpha_fast", "alpha_basic", "alpha_slow"]
BC6H = ["veryfast", "fast", "basic", "slow", "veryslow"]


def _ensure_ref(name):
    """Path of an oracle/_ref artefact; built on demand where the reference tree exists, skipped elsewhere."""
    p = os.path.join(REF, name)
    if not os.path.exists(p):
        if not os.path.exists(REFERENCE_TREE):
            if os.path.exists("/dev/kfd") and os.environ.get("ITW_ALLOW_NO_REF") != "1":
                # a GPU box: the prebuilt checkers are part of what must travel (no silent downgrade)
                pytest.fail(f"oracle/_ref/{name} did not travel to this GPU box (build in the container: make -C oracle/ref_build; "
                            "ITW_ALLOW_NO_REF=1 to run without)")
            pytest.skip(f"oracle/_ref/{name} not prebuilt and /root/reference absent")
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "ref_build")], check=True)
    assert os.path.exists(p), f"oracle/ref_build did not produce {name}"
    return p


def _golden_presets():
    b = open(os.path.join(ROOT, "tests", "golden", "ref_presets.bin"), "rb").read()
    assert len(b) == 10 * 64 + 5 * 16
    out = {("bc7", n): b[64 * i:64 * i + 64] for i, n in enumerate(BC7)}
    out.update({("bc6h", n): b[640 + 16 * i:640 + 16 * i + 16] for i, n in enumerate(BC6H)})
    return out


def _fill_sentinel(fn, size):
    buf = (C.c_uint8 * size)(*([0xA5] * size))
    fn(C.cast(buf, C.c_void_p))
    return bytes(buf)


# ---------------------------------------------------------------- presets: reference TU == golden == product == oracle

def test_golden_presets_are_what_the_reference_tu_produces_now():
    """tests/golden/ref_presets.bin is regenerated from the reference's own ispc_texcomp.cpp and must not have drifted."""
    exe = _ensure_ref("ref_header_caller_cpu")
    out = os.path.join(REF, "presets_now.bin")
    subprocess.run([exe, "profiles", out], check=True, timeout=60)
    assert open(out, "rb").read() == open(os.path.join(ROOT, "tests", "golden", "ref_presets.bin"), "rb").read()


def test_reference_tu_library_fills_presets_like_the_golden():
    L = C.CDLL(_ensure_ref("libispc_texcomp_ref.so"))
    g = _golden_presets()
    for n in BC7:
        assert _fill_sentinel(getattr(L, "GetProfile_" + n), 64) == g[("bc7", n)], n
    for n in BC6H:
        assert _fill_sentinel(getattr(L, "GetProfile_bc6h_" + n), 16) == g[("bc6h", n)], n


@pytest.mark.parametrize("name", BC7)
def test_product_bc7_preset_bytes_equal_the_reference_tu(itw, name):
    """All 64 bytes over 0xA5 storage: written fields, untouched padding, and refineIterations[7] (which the RGB presets
    leave unwritten, ispc_texcomp.cpp:20-189) must match the reference TU byte for byte."""
    fn = getattr(itw.lib(), "GetProfile_" + name)
    fn.restype = None
    assert _fill_sentinel(fn, 64).hex() == _golden_presets()[("bc7", name)].hex()


@pytest.mark.parametrize("name", BC6H)
def test_product_bc6h_preset_bytes_equal_the_reference_tu(itw, name):
    fn = getattr(itw.lib(), "GetProfile_bc6h_" + name)
    fn.restype = None
    assert _fill_sentinel(fn, 16).hex() == _golden_presets()[("bc6h", name)].hex()


def test_oracle_presets_equal_the_reference_tu(oracle):
    L = oracle.lib()
    g = _golden_presets()
    for kind, names, size, fn in (("bc7", BC7, 64, L.oracle_GetProfile_bc7), ("bc6h", BC6H, 16, L.oracle_GetProfile_bc6h)):
        for n in names:
            buf = (C.c_uint8 * size)(*([0xA5] * size))
            assert fn(n.encode(), C.cast(buf, C.c_void_p)) == 0
            assert bytes(buf).hex() == g[(kind, n)].hex(), (kind, n)


# ------------------------------------------------ the reference's dispatch layer (win32Threads.cpp) over the oracle, CPU

def _surface(fmt, h, w):
    from itw_amd import surfaces
    return surfaces.hdr_smooth(h, w) if fmt == "bc6h" else surfaces.ldr_smooth(h, w)


def _run_threads_caller(exe, mode, tramp, img, tmp_path, workers, whole=False):
    h, w = img.shape[:2]
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    img.tofile(raw)
    env = dict(os.environ, ITW_REF_THREADS=str(workers))
    r = subprocess.run([exe, mode, tramp, str(w), str(h), str(raw), str(out)] + (["whole"] if whole else []),
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr
    return np.fromfile(out, dtype=np.uint8), r.stdout


@pytest.mark.parametrize("tramp,fmt,prof,h,w,workers", [
    ("BC1", "bc1", None, 600, 512, 7),              # two plugin slices, band starts 0,40,84,... (win32Threads.cpp:223)
    ("BC3", "bc3", None, 72, 64, 64),               # more workers than block rows: empty bands
    ("BC7_veryfast", "bc7", "veryfast", 100, 64, 3),
    ("BC6H_fast", "bc6h", "fast", 64, 48, 5),
])
def test_reference_dispatch_over_reference_tu_equals_one_oracle_call(oracle, tmp_path, tramp, fmt, prof, h, w, workers):
    """Reference CompressImageMT band rule + reference presets + restated kernel == one whole-surface oracle call: the
    reference's own splitting never changes a byte (blocks are independent), which is what lets the GPU library see
    whole surfaces instead."""
    exe = _ensure_ref("ref_threads_caller_cpu")
    img = _surface(fmt, h, w)
    bpb = 8 if fmt == "bc1" else 16
    want = oracle.encode(fmt, img, prof).reshape(-1)
    for mode in ("mt", "st"):
        got, _ = _run_threads_caller(exe, mode, tramp, img, tmp_path, workers)
        assert first_mismatch(got, want, bpb) is None, (mode, first_mismatch(got, want, bpb))


@pytest.mark.parametrize("tramp,fmt,prof,h,w,workers", [
    ("BC3", "bc3", None, 200, 128, 5),
    ("BC7_alpha_basic", "bc7", "alpha_basic", 100, 64, 3),
    ("BC7_slow", "bc7", "slow", 64, 64, 64),
    ("BC6H_slow", "bc6h", "slow", 64, 48, 4),
])
def test_the_all_reference_stack_equals_the_oracle(oracle, tmp_path, tramp, fmt, prof, h, w, workers):
    """ref_threads_caller_ref: the reference's win32Threads.cpp (trampolines, CompressImageMT/ST) over its ispc_texcomp.cpp
    (presets, ABI wrappers) over its kernel.ispc built as a scalar program -- no line of encoder code in that executable
    is ours -- emits the bytes of one whole-surface oracle call."""
    exe = _ensure_ref("ref_threads_caller_ref")
    img = _surface(fmt, h, w)
    want = oracle.encode(fmt, img, prof).reshape(-1)
    for mode in ("mt", "st"):
        got, _ = _run_threads_caller(exe, mode, tramp, img, tmp_path, workers)
        assert first_mismatch(got, want, 8 if fmt == "bc1" else 16) is None, (mode, first_mismatch(got, want, 8 if fmt == "bc1" else 16))


# ---------------------------------------------------------------------------- the same reference-built callers on the GPU

@pytest.mark.gpu
def test_caller_compiled_against_the_reference_header_gets_reference_presets_from_the_product(gpu):
    exe = _ensure_ref("ref_header_caller_gpu")
    out = os.path.join(REF, "presets_product.bin")
    subprocess.run([exe, "profiles", out], check=True, timeout=120)
    assert open(out, "rb").read() == open(os.path.join(ROOT, "tests", "golden", "ref_presets.bin"), "rb").read()


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,prof,h,w", [("bc1", "-", 128, 256), ("bc3", "-", 64, 64), ("bc7", "slow", 64, 128),
                                          ("bc7", "alpha_basic", 96, 64), ("bc6h", "slow", 32, 64), ("bc6h", "fast", 64, 32)])
def test_caller_compiled_against_the_reference_header_links_the_product_and_matches_the_oracle(gpu, oracle, tmp_path, fmt, prof, h, w):
    exe = _ensure_ref("ref_header_caller_gpu")
    img = _surface(fmt, h, w)
    raw, out = tmp_path / "in.raw", tmp_path / "out.bin"
    img.tofile(raw)
    r = subprocess.run([exe, "encode", fmt, prof, str(w), str(h), str(raw), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    want = oracle.encode(fmt, img, None if prof == "-" else prof).reshape(-1)
    got = np.fromfile(out, dtype=np.uint8)
    assert first_mismatch(got, want, 8 if fmt == "bc1" else 16) is None, first_mismatch(got, want, 8 if fmt == "bc1" else 16)


@pytest.mark.gpu
@pytest.mark.parametrize("tramp,fmt,prof,h,w,workers", [
    ("BC1", "bc1", None, 1024, 1024, 16),
    ("BC3", "bc3", None, 600, 512, 7),
    ("BC7_basic", "bc7", "basic", 1024, 512, 64),            # 2 slices x 64 bands of 8 rows: the worst legacy granularity
    ("BC7_alpha_veryfast", "bc7", "alpha_veryfast", 512, 512, 8),
    ("BC6H_slow", "bc6h", "slow", 512, 512, 12),
])
def test_reference_dispatch_code_drives_the_product_unchanged(gpu, oracle, tmp_path, tramp, fmt, prof, h, w, workers):
    """win32Threads.cpp as shipped (slice loop restated from IntelPlugin.cpp:851-879, band-per-thread CompressImageMT,
    trampolines with stack settings) calling libispc_texcomp.so from `workers` concurrent host threads."""
    exe = _ensure_ref("ref_threads_caller_gpu")
    img = _surface(fmt, h, w)
    bpb = 8 if fmt == "bc1" else 16
    want = oracle.encode(fmt, img, prof).reshape(-1)
    for mode in ("mt", "st"):
        got, _ = _run_threads_caller(exe, mode, tramp, img, tmp_path, workers)
        assert first_mismatch(got, want, bpb) is None, (mode, first_mismatch(got, want, bpb))


def test_prepass_oracle_equals_the_reference_scalar_conversions(oracle):
    """oracle/prepass.c (the checker of csrc/convert.hip) was a restatement nothing reference-held pinned.
    oracle/_ref/libintelplugin_convert_ref.so is IntelPlugin.h:31-96 compiled from where it lies (F32toF16 / FloatToByte /
    ConvertTo8Bit x 3 / ConvertTo16Bit x 3; typedef shim for the Photoshop SDK's scalar types, DirectXMath half conversions
    from dxmath_stub/).  Exhaustive over the 8- and 16-bit sources, a dense sample incl. every special class for the 32-bit
    ones -- byte for byte, the gamma path too (same libm on the same host)."""
    L = C.CDLL(_ensure_ref("libintelplugin_convert_ref.so"))
    L.ref_convert8_from8.argtypes = [C.c_uint8]; L.ref_convert8_from8.restype = C.c_uint8
    L.ref_convert8_from16.argtypes = [C.c_uint16]; L.ref_convert8_from16.restype = C.c_uint8
    L.ref_convert8_from32.argtypes = [C.c_float, C.c_int]; L.ref_convert8_from32.restype = C.c_uint8
    L.ref_convert16_from8.argtypes = [C.c_uint8]; L.ref_convert16_from8.restype = C.c_uint16
    L.ref_convert16_from16.argtypes = [C.c_uint16]; L.ref_convert16_from16.restype = C.c_uint16
    L.ref_convert16_from32.argtypes = [C.c_float]; L.ref_convert16_from32.restype = C.c_uint16
    O = oracle.lib()

    def via_oracle8(src, depth, gamma=0):
        out = np.zeros((src.size, 4), np.uint8)
        O.oracle_convert_rgba8(src.ctypes.data_as(C.c_void_p), depth, 1, 0, gamma, src.size, 1, out.ctypes.data_as(C.c_void_p))
        assert (out[:, 1:3] == 0).all() and (out[:, 3] == 255).all()
        return out[:, 0]

    def via_oracle16(src, depth):
        out = np.zeros((src.size, 4), np.uint16)
        O.oracle_convert_rgba16f(src.ctypes.data_as(C.c_void_p), depth, 1, 0, src.size, 1, out.ctypes.data_as(C.c_void_p))
        assert (out[:, 1:3] == 0).all() and (out[:, 3] == 0x3c00).all()
        return out[:, 0]

    b = np.arange(256, dtype=np.uint8)
    assert np.array_equal(via_oracle8(b, 8), np.array([L.ref_convert8_from8(int(v)) for v in b], np.uint8))
    assert np.array_equal(via_oracle16(b, 8), np.array([L.ref_convert16_from8(int(v)) for v in b], np.uint16))
    w = np.arange(65536, dtype=np.uint16)                               # Photoshop's 0..32768 range and everything above it
    assert np.array_equal(via_oracle8(w, 16), np.array([L.ref_convert8_from16(int(v)) for v in w], np.uint8))
    assert np.array_equal(via_oracle16(w, 16), np.array([L.ref_convert16_from16(int(v)) for v in w], np.uint16))
    rng = np.random.default_rng(8)
    f = np.concatenate([np.linspace(-0.25, 1.25, 20001), rng.uniform(0, 1, 20000), np.exp2(rng.uniform(-30, 15.9, 20000)),
                        -np.exp2(rng.uniform(-30, 15.9, 2000)), [0.0, -0.0, 1.0, 65504.0, 6.1e-5, 5.96e-8, 2.9e-8, 1e-10]]).astype(np.float32)
    for gamma in (0, 1):
        src = f[f >= 0] if gamma else f                                 # pow of a negative base is NaN: its byte cast is unspecified
        assert np.array_equal(via_oracle8(src, 32, gamma), np.array([L.ref_convert8_from32(float(v), gamma) for v in src], np.uint8)), gamma
    fin = f[np.abs(f) <= 65504.0]                                       # beyond: DirectXMath releases disagree (oracle/prepass.c header)
    assert np.array_equal(via_oracle16(fin, 32), np.array([L.ref_convert16_from32(float(v)) for v in fin], np.uint16))


def test_committed_gamma_thresholds_are_what_the_reference_function_gives_now():
    """csrc/gamma_thresholds.h is generated from the reference's own ConvertTo8Bit(double, true) compiled where it lies
    (tools/gen_gamma_thresholds.py); where that build exists the committed table must be what the generator produces."""
    import subprocess
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libintelplugin_convert_ref.so")):
        pytest.skip("oracle/_ref not built here")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gen_gamma_thresholds.py"), "--check"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, "csrc/gamma_thresholds.h differs from what tools/gen_gamma_thresholds.py generates: " + r.stderr[-500:]
