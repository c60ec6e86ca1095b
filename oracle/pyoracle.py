"""TEST INFRASTRUCTURE -- ctypes face of oracle/liboracle_bcn.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
module.  It is the checker, never the product: the product library
(libispc_texcomp.so) has no dependency on anything in oracle/.

Parity status: no golden outputs exist upstream and ispc is not installable here; the
restatement is pinned by the reference's own kernel.ispc built as a scalar program
(oracle/pyref.py, tests/test_reference_kernel_source.py); the ispc compiler / stdlib
semantics stay assumed (oracle/x86_math.h).
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle_bcn.so")


class Surface(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32)]


class Bc7Settings(C.Structure):
    _fields_ = [("mode_selection", C.c_uint8 * 4), ("refineIterations", C.c_int32 * 8),
                ("skip_mode2", C.c_uint8), ("fastSkipTreshold_mode1", C.c_int32),
                ("fastSkipTreshold_mode3", C.c_int32), ("fastSkipTreshold_mode7", C.c_int32),
                ("mode45_channel0", C.c_int32), ("refineIterations_channel", C.c_int32),
                ("channels", C.c_int32)]


class Bc6hSettings(C.Structure):
    _fields_ = [("slow_mode", C.c_uint8), ("fast_mode", C.c_uint8), ("refineIterations_1p", C.c_int32),
                ("refineIterations_2p", C.c_int32), ("fastSkipTreshold", C.c_int32)]


def build(force=False):
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".c", ".h"))]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.run(["make", "-C", _HERE, "-B" if force else "-s"], check=True,
                   stdout=subprocess.DEVNULL if not force else None)
    return _LIB_PATH


_lib = None


def _bind(path):
        L = C.CDLL(path)
        for n in ("oracle_rcp", "oracle_rsqrt", "oracle_rcpps", "oracle_rsqrtps"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        L.oracle_f2i.restype = C.c_int32
        L.oracle_f2i.argtypes = [C.c_float]
        for n in ("oracle_CompressBlocksBC1", "oracle_CompressBlocksBC3"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, n).restype = None
        for n in ("oracle_CompressBlocksBC7", "oracle_CompressBlocksBC6H"):
            if hasattr(L, n):
                getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
                getattr(L, n).restype = None
        for n in ("oracle_GetProfile_bc7", "oracle_GetProfile_bc6h"):
            if hasattr(L, n):
                getattr(L, n).argtypes = [C.c_char_p, C.c_void_p]
                getattr(L, n).restype = C.c_int
        for n in ("oracle_decode_bc1", "oracle_decode_bc3", "oracle_decode_bc7", "oracle_decode_bc6h", "oracle_decode_bc4_rgba8", "oracle_decode_bc5_rgba8"):
            if hasattr(L, n):
                getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
                getattr(L, n).restype = C.c_int
        for n in ("oracle_CompressBlocksBC4", "oracle_CompressBlocksBC5", "oracle_bc4_block", "oracle_decode_bc4_float"):
            getattr(L, n).argtypes = [C.c_void_p, C.c_void_p]
            getattr(L, n).restype = None
        L.oracle_bc4_find_closest_row.argtypes = [C.c_int, C.c_int, C.c_void_p]
        L.oracle_bc4_find_closest_row.restype = None
        return L


def lib():
    global _lib
    if _override is not None:
        return _override
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = _bind(_LIB_PATH)
    return _lib


# Arithmetic-model variants of the oracle (x86_math.h switches; `make -C oracle variants`).  Study material only
# (tools/arith_sensitivity.py, tests/test_arith_models.py): parity is always against the default model.
VARIANTS = ("div1158rcp", "ieee", "fma", "ieee_fma", "reassoc")
_override = None
_variant_libs = {}


class variant:
    """`with pyoracle.variant("ieee"): pyoracle.encode(...)` -- run the oracle under another arithmetic model."""

    def __init__(self, name):
        assert name in VARIANTS, name
        self.name = name

    def __enter__(self):
        global _override
        if self.name not in _variant_libs:
            path = os.path.join(_HERE, "_variants", f"liboracle_bcn_{self.name}.so")
            # make decides: a variant older than the sources (a function added since) is rebuilt, like the main library
            subprocess.run(["make", "-C", _HERE, "-s", "variants"], check=True)
            _variant_libs[self.name] = _bind(path)
        self._saved, _override = _override, _variant_libs[self.name]
        return self

    def __exit__(self, *exc):
        global _override
        _override = self._saved


def has(symbol):
    return hasattr(lib(), symbol)


def _surface(arr):
    assert arr.flags["C_CONTIGUOUS"] and arr.ndim == 3
    h, w = arr.shape[:2]
    return Surface(arr.ctypes.data, w, h, arr.strides[0])


def bc7_profile(name):
    s = Bc7Settings()
    rc = lib().oracle_GetProfile_bc7(name.encode(), C.byref(s))
    if rc != 0:
        raise KeyError(name)
    return s


def bc6h_profile(name):
    s = Bc6hSettings()
    rc = lib().oracle_GetProfile_bc6h(name.encode(), C.byref(s))
    if rc != 0:
        raise KeyError(name)
    return s


def encode(fmt, img, profile=None, rows=None):
    """img: (H, W, 4) uint8 (bc1/bc3/bc7) or uint16 half bits (bc6h).  Returns bytes as a uint8 array.
    rows=(y0, y1) restricts to a texel-row band (multiples of 4)."""
    img = np.ascontiguousarray(img)
    if rows is not None:
        img = img[rows[0]:rows[1]]
    if fmt in ("bc4", "bc5"):
        return encode_bc45(fmt, img)
    h, w = img.shape[:2]
    # kernel.ispc:157 places block row yy at byte yy*width*data_size: only for width % 4 == 0 is that the tight
    # pitch (width/4)*bytes_per_block.  Other widths are outside the reference's contract (ispc_texcomp.h:95).
    assert w % 4 == 0, "oracle: width must be a multiple of 4 (reference output pitch is width*data_size bytes)"
    bpb = 8 if fmt == "bc1" else 16
    out = np.zeros((h // 4) * (w // 4) * bpb, dtype=np.uint8)
    s = _surface(img)
    L = lib()
    dst = out.ctypes.data_as(C.c_void_p)
    if fmt == "bc1":
        assert img.dtype == np.uint8
        L.oracle_CompressBlocksBC1(C.byref(s), dst)
    elif fmt == "bc3":
        assert img.dtype == np.uint8
        L.oracle_CompressBlocksBC3(C.byref(s), dst)
    elif fmt == "bc7":
        assert img.dtype == np.uint8
        st = profile if isinstance(profile, Bc7Settings) else bc7_profile(profile or "slow")
        L.oracle_CompressBlocksBC7(C.byref(s), dst, C.byref(st))
    elif fmt == "bc6h":
        assert img.dtype == np.uint16
        st = profile if isinstance(profile, Bc6hSettings) else bc6h_profile(profile or "slow")
        L.oracle_CompressBlocksBC6H(C.byref(s), dst, C.byref(st))
    else:
        raise ValueError(fmt)
    return out


def encode_bc45(fmt, img):
    """BC4/BC5 (DirectXTex path): img (H, W, 4) uint8, any H, W >= 1 -> ceil(W/4) x ceil(H/4) blocks."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 4
    h, w = img.shape[:2]
    bpb = 8 if fmt == "bc4" else 16
    out = np.zeros(((h + 3) // 4) * ((w + 3) // 4) * bpb, dtype=np.uint8)
    s = _surface(img)
    fn = lib().oracle_CompressBlocksBC4 if fmt == "bc4" else lib().oracle_CompressBlocksBC5
    fn(C.byref(s), out.ctypes.data_as(C.c_void_p))
    return out


def bc4_block(texels):
    """One channel of one block: 16 float32 texels in [0, 1] -> 8 bytes."""
    t = np.ascontiguousarray(texels, dtype=np.float32).reshape(16)
    out = np.zeros(8, dtype=np.uint8)
    lib().oracle_bc4_block(t.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    return out


def bc4_find_closest_table():
    """FindClosestUNORM over its whole domain: uint8 [256 r0][256 r1][256 texel codes]."""
    out = np.zeros((256, 256, 256), dtype=np.uint8)
    fn = lib().oracle_bc4_find_closest_row
    for r0 in range(256):
        for r1 in range(256):
            fn(r0, r1, out[r0, r1].ctypes.data_as(C.c_void_p))
    return out


def decode_bc45(fmt, blocks, width, height):
    """DirectXTex's float decode of a BC4/BC5 stream -> (ceil4(H), ceil4(W), channels) float32."""
    nch = 1 if fmt == "bc4" else 2
    bx, by = (width + 3) // 4, (height + 3) // 4
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(by, bx, nch, 8)
    out = np.zeros((by, bx, nch, 16), dtype=np.float32)
    L = lib()
    for y in range(by):
        for x in range(bx):
            for c in range(nch):
                L.oracle_decode_bc4_float(blocks[y, x, c].ctypes.data_as(C.c_void_p), out[y, x, c].ctypes.data_as(C.c_void_p))
    return out.reshape(by, bx, nch, 4, 4).transpose(0, 3, 1, 4, 2).reshape(by * 4, bx * 4, nch)


def usable_cores():
    """Hardware threads this process may actually use: min(cpu_count, affinity mask, cgroup CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, -(-int(quota) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def encode_mt(fmt, img, profile=None, threads=None):
    """Row-band threaded encode with the reference's band rule (win32Threads.cpp:217-231):
    linesPerThread = ceil(h/n), y_start = lines*i/4*4.  ctypes releases the GIL."""
    from concurrent.futures import ThreadPoolExecutor
    h, w = img.shape[:2]
    n = threads or usable_cores()
    lines = (h + n - 1) // n
    bands = []
    for i in range(n):
        y0 = (lines * i) // 4 * 4
        y1 = h if i == n - 1 else (lines * (i + 1)) // 4 * 4
        y1 = min(y1, h)
        if y1 > y0:
            bands.append((y0, y1))
    with ThreadPoolExecutor(max_workers=len(bands)) as ex:
        parts = list(ex.map(lambda b: encode(fmt, img, profile, rows=b), bands))
    return np.concatenate(parts)


def decode(fmt, blocks, width, height):
    """From-spec decode of a tightly packed block stream -> (H, W, 4) uint8, or (H, W, 3) uint16 for bc6h."""
    L = lib()
    bpb = 8 if fmt in ("bc1", "bc4") else 16
    bx, by = width // 4, height // 4
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(by * bx, bpb)
    if fmt == "bc6h":
        out = np.zeros((by * 4, bx * 4, 3), dtype=np.uint16)
        tmp = (C.c_uint16 * 48)()
        modes = np.zeros(by * bx, dtype=np.int32)
        for i in range(by * bx):
            modes[i] = L.oracle_decode_bc6h(blocks[i].ctypes.data_as(C.c_void_p), tmp)
            t = np.frombuffer(tmp, dtype=np.uint16).reshape(3, 4, 4)   # planar [ch][y][x]
            y, x = divmod(i, bx)
            out[y * 4:y * 4 + 4, x * 4:x * 4 + 4, :] = np.transpose(t, (1, 2, 0))
        return out, modes
    out = np.zeros((by * 4, bx * 4, 4), dtype=np.uint8)
    tmp = (C.c_uint8 * 64)()
    modes = np.zeros(by * bx, dtype=np.int32)
    fn = {"bc1": L.oracle_decode_bc1, "bc3": L.oracle_decode_bc3, "bc7": L.oracle_decode_bc7,
          "bc4": L.oracle_decode_bc4_rgba8, "bc5": L.oracle_decode_bc5_rgba8}[fmt]
    fn.restype = C.c_int
    for i in range(by * bx):
        r = fn(blocks[i].ctypes.data_as(C.c_void_p), tmp)
        modes[i] = r if fmt == "bc7" else 0
        t = np.frombuffer(tmp, dtype=np.uint8).reshape(4, 4, 4)        # [y][x][rgba]
        y, x = divmod(i, bx)
        out[y * 4:y * 4 + 4, x * 4:x * 4 + 4, :] = t
    return out, modes
