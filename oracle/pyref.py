"""TEST INFRASTRUCTURE -- ctypes face of oracle/_ref/libispc_texcomp_ref_full.so: the reference's OWN library built
without the ispc compiler (oracle/ref_build/Makefile): ispc_texcomp.cpp unmodified + kernel.ispc compiled as one scalar
program instance (oracle/ref_build/ispc_as_cpp/).  Same C ABI as the reference (ispc_texcomp.h:67-107).

Only tests/ and tools/ import this.  It is a checker of the checker: it pins oracle/*.c's reading of kernel.ispc by
the reference's own source, under the same assumed ispc compiler / stdlib semantics (ispc_prelude.h S1-S10)."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import pyoracle

_HERE = os.path.dirname(os.path.abspath(__file__))
VARIANTS = {"": "libispc_texcomp_ref_full.so", "div1158rcp": "libispc_texcomp_ref_full_div1158rcp.so",
            "ieee": "libispc_texcomp_ref_full_ieee.so"}
_libs = {}


def path(variant=""):
    return os.path.join(_HERE, "_ref", VARIANTS[variant])


def available(variant=""):
    return os.path.exists(path(variant))


def lib(variant=""):
    if variant not in _libs:
        _libs[variant] = C.CDLL(path(variant), mode=os.RTLD_LOCAL)
    return _libs[variant]


def bc7_profile(name, variant=""):
    s = pyoracle.Bc7Settings()
    getattr(lib(variant), "GetProfile_" + name)(C.byref(s))
    return s


def bc6h_profile(name, variant=""):
    s = pyoracle.Bc6hSettings()
    getattr(lib(variant), "GetProfile_bc6h_" + name)(C.byref(s))
    return s


def encode(fmt, img, profile=None, variant=""):
    """Block stream of `img` ((H, W, 4) uint8, or uint16 / int16 half bits for bc6h) from the reference's kernel source.
    `profile`: preset name or a ready settings struct (pyoracle.Bc7Settings / Bc6hSettings)."""
    L = lib(variant)
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    surf = pyoracle.Surface(img.ctypes.data, w, h, img.strides[0])
    out = np.zeros((h // 4) * (w // 4) * (8 if fmt == "bc1" else 16), np.uint8)
    dst = out.ctypes.data_as(C.c_void_p)
    if fmt == "bc1":
        L.CompressBlocksBC1(C.byref(surf), dst)
    elif fmt == "bc3":
        L.CompressBlocksBC3(C.byref(surf), dst)
    elif fmt == "bc7":
        s = bc7_profile(profile, variant) if isinstance(profile, str) else profile
        L.CompressBlocksBC7(C.byref(surf), dst, C.byref(s))
    elif fmt == "bc6h":
        s = bc6h_profile(profile, variant) if isinstance(profile, str) else profile
        L.CompressBlocksBC6H(C.byref(surf), dst, C.byref(s))
    else:
        raise ValueError(fmt)
    return out


def encode_mt(fmt, img, profile=None, threads=None, variant=""):
    """encode() over row bands on several threads (ctypes releases the GIL; blocks are independent)."""
    h = img.shape[0]
    n = max(1, min(threads or pyoracle.usable_cores(), h // 4))
    rows = [(h // 4) * i // n * 4 for i in range(n + 1)]
    with ThreadPoolExecutor(n) as ex:
        parts = list(ex.map(lambda i: encode(fmt, img[rows[i]:rows[i + 1]], profile, variant), range(n)))
    return np.concatenate(parts)
