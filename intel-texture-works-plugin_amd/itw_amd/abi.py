"""ctypes binding of the drop-in C ABI (include/ispc_texcomp.h, include/itw_amd.h)."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.normpath(os.path.join(_PKG, "..", "lib", "libispc_texcomp.so"))
_TEST_LIB = os.path.normpath(os.path.join(_PKG, "..", "lib", "libispc_texcomp_test.so"))

BYTES_PER_BLOCK = {"bc1": 8, "bc3": 16, "bc7": 16, "bc6h": 16, "bc4": 8, "bc5": 16}
KEEPS_PARTIAL_BLOCKS = ("bc4", "bc5")     # DirectXTex formats: ceil(w/4) x ceil(h/4) blocks (include/itw_bc45.h)
BC7_PROFILES = ("ultrafast", "veryfast", "fast", "basic", "slow",
                "alpha_ultrafast", "alpha_veryfast", "alpha_fast", "alpha_basic", "alpha_slow")
BC6H_PROFILES = ("veryfast", "fast", "basic", "slow", "veryslow")

# every symbol include/*.h declares
EXPORTED_SYMBOLS = tuple(
    ["CompressBlocksBC1", "CompressBlocksBC3", "CompressBlocksBC6H", "CompressBlocksBC7"]
    + ["GetProfile_" + p for p in BC7_PROFILES] + ["GetProfile_bc6h_" + p for p in BC6H_PROFILES]
    + ["itwSetStream", "itwGetStream", "itwAvailable", "itwSetErrorMode", "itwLastError", "itwClearError", "itwSetBc7Path", "itwSetBc7Pilot",
       "itwDeviceInfo", "itwVersion", "itwBandForPart", "itwBandForPartEx"]
    # include/itw_dispatch.h: the reference's dispatch layer (win32Threads.h), slice loop, pad pre-pass
    + ["GetProcessorCount", "InitWin32Threads", "DestroyThreads", "GetBytesPerBlock", "CompressImageMT", "CompressImageST",
       "CompressImageBC1", "CompressImageBC3", "CompressImageBC4", "CompressImageBC5"]
    + ["CompressImageBC7_" + p for p in BC7_PROFILES] + ["CompressImageBC6H_" + p for p in BC6H_PROFILES]
    + ["itwCompressImageSliced", "itwCompressImageSlicedEx", "itwSetSliceWindow", "itwSliceWindow", "itwSliceWindowFor", "itwPadToMultipleOf4", "itwFreeSurface", "itwPadToMultipleOf4Device",
       "itwConvertToRGBA8Device", "itwConvertToRGBA16FDevice"]
    # include/itw_multigpu.h: one surface over all GPUs, one process
    + ["itwMultiGpuRanks", "itwMultiGpuTransport", "itwMultiGpuPeerLinks", "itwCompressImageMultiGPU", "itwCompressImageMultiGPUEx", "itwCompressImageMultiGPUBands",
       "itwMultiGpuSetInterleave", "itwMultiGpuPieces"]
    # include/itw_bc45.h: the DirectXTex formats of the plugin
    + ["CompressBlocksBC4", "CompressBlocksBC5", "itwWarmupBC45"]
    # include/itw_decode.h: device decoders
    + ["itwDecodeBlocks"]
    # include/itw_dds.h: DDS container
    + ["itwDdsLevelBytes", "itwDdsHeaderBytes", "itwDdsFileBytes", "itwDdsWriteHeader", "itwDdsReadHeader", "itwDdsWriteFile"])
# include/itw_test_hooks.h: exported by libispc_texcomp_test.so only (the same sources built with -DITW_TEST_HOOKS), never by the product
TEST_HOOK_SYMBOLS = ("itwTestRcp", "itwTestRsqrt", "itwTestF2I", "itwTestBc7TwoSubsetBounds", "itwTestBc45IndexTable", "itwMultiGpuTestInjectFailure")




class RgbaSurface(C.Structure):
    """struct rgba_surface (ispc_texcomp.h:19-25): 24 bytes."""
    _fields_ = [("ptr", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_int32)]


class Bc7Settings(C.Structure):
    """struct bc7_enc_settings (ispc_texcomp.h:27-41): 64 bytes."""
    _fields_ = [("mode_selection", C.c_bool * 4), ("refineIterations", C.c_int * 8),
                ("skip_mode2", C.c_bool), ("fastSkipTreshold_mode1", C.c_int),
                ("fastSkipTreshold_mode3", C.c_int), ("fastSkipTreshold_mode7", C.c_int),
                ("mode45_channel0", C.c_int), ("refineIterations_channel", C.c_int),
                ("channels", C.c_int)]


class Bc6hSettings(C.Structure):
    """struct bc6h_enc_settings (ispc_texcomp.h:43-50): 16 bytes."""
    _fields_ = [("slow_mode", C.c_bool), ("fast_mode", C.c_bool), ("refineIterations_1p", C.c_int),
                ("refineIterations_2p", C.c_int), ("fastSkipTreshold", C.c_int)]


assert C.sizeof(RgbaSurface) == 24 and C.sizeof(Bc7Settings) == 64 and C.sizeof(Bc6hSettings) == 16

DXGI_FORMAT = {"bc1": 71, "bc1_srgb": 72, "bc3": 77, "bc3_srgb": 78, "bc4": 80, "bc5": 83, "bc6h": 95, "bc6h_sf16": 96, "bc7": 98, "bc7_srgb": 99}


class DdsDesc(C.Structure):
    """struct ItwDdsDesc (itw_dds.h)."""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("mip_levels", C.c_uint32), ("dxgi_format", C.c_uint32),
                ("is_cubemap", C.c_uint32), ("array_size", C.c_uint32)]


class MultiGpuRankStats(C.Structure):
    """struct itw_multigpu_rank_stats (itw_multigpu.h)."""
    _fields_ = [("rank", C.c_int32), ("device", C.c_int32), ("block_row0", C.c_int32), ("block_rows", C.c_int32),
                ("upload_ms", C.c_float), ("encode_ms", C.c_float), ("gather_ms", C.c_float), ("span_ms", C.c_float)]


class MultiGpuStats(C.Structure):
    """struct itw_multigpu_stats (itw_multigpu.h)."""
    _fields_ = [("ranks", C.c_int32), ("devices", C.c_int32), ("peer_links", C.c_int32), ("rccl_ranks", C.c_int32),
                ("watchdog_fired", C.c_int32), ("resident_bands", C.c_int32), ("wall_ms", C.c_float), ("posted_ms", C.c_float),
                ("transport", C.c_char * 8), ("transport_note", C.c_char * 96), ("rank", MultiGpuRankStats * 64),
                ("interleave", C.c_int32)]          # appended after round 4: the offsets above never move

    def as_dict(self):
        n = max(0, min(int(self.ranks), 64))
        return {"ranks": int(self.ranks), "devices": int(self.devices), "peer_links": int(self.peer_links), "rccl_ranks": int(self.rccl_ranks),
                "watchdog_fired": bool(self.watchdog_fired), "resident_bands": bool(self.resident_bands), "interleave": int(self.interleave),
                "wall_ms": round(float(self.wall_ms), 4), "posted_ms": round(float(self.posted_ms), 4),
                "transport": self.transport.decode(), "transport_note": self.transport_note.decode(),
                "per_rank": [{"rank": int(r.rank), "device": int(r.device), "block_row0": int(r.block_row0), "block_rows": int(r.block_rows),
                              "upload_ms": round(float(r.upload_ms), 4), "encode_ms": round(float(r.encode_ms), 4),
                              "gather_ms": round(float(r.gather_ms), 4), "span_ms": round(float(r.span_ms), 4)} for r in self.rank[:n]]}


COMPRESSION_FUNC = C.CFUNCTYPE(None, C.POINTER(RgbaSurface), C.c_void_p)
PROGRESS_FUNC = C.CFUNCTYPE(C.c_bool, C.c_int, C.c_int, C.c_void_p)

_lib = None
_test_lib = None


def lib_path():
    return _LIB


def lib():
    """Load libispc_texcomp.so (the product); raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        _lib = _load(_LIB, hooks=False)
    return _lib


def test_lib():
    """Load libispc_texcomp_test.so: the product's sources built with -DITW_TEST_HOOKS, which adds the entry points of
    include/itw_test_hooks.h.  A second, independent instance of the library in the process; tests/ use it for the hook calls only."""
    global _test_lib
    if _test_lib is None:
        _test_lib = _load(_TEST_LIB, hooks=True)
    return _test_lib


def _load(path, hooks):
    if True:
        if not os.path.exists(path):
            raise RuntimeError(
                f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C intel-texture-works-plugin_amd/csrc`.  There is no CPU fallback.")
        L = C.CDLL(path, mode=C.RTLD_LOCAL if hooks else C.RTLD_GLOBAL)
        L.CompressBlocksBC1.argtypes = [C.POINTER(RgbaSurface), C.c_void_p]
        L.CompressBlocksBC4.argtypes = [C.POINTER(RgbaSurface), C.c_void_p]
        L.CompressBlocksBC5.argtypes = [C.POINTER(RgbaSurface), C.c_void_p]
        L.CompressBlocksBC4.restype = None
        L.itwWarmupBC45.restype = None
        L.CompressBlocksBC5.restype = None
        L.CompressBlocksBC3.argtypes = [C.POINTER(RgbaSurface), C.c_void_p]
        L.CompressBlocksBC7.argtypes = [C.POINTER(RgbaSurface), C.c_void_p, C.POINTER(Bc7Settings)]
        L.CompressBlocksBC6H.argtypes = [C.POINTER(RgbaSurface), C.c_void_p, C.POINTER(Bc6hSettings)]
        for n in ("CompressBlocksBC1", "CompressBlocksBC3", "CompressBlocksBC7", "CompressBlocksBC6H"):
            getattr(L, n).restype = None
        L.itwSetStream.argtypes = [C.c_void_p]
        L.itwSetStream.restype = None
        L.itwGetStream.restype = C.c_void_p
        L.itwDeviceInfo.restype = C.c_char_p
        L.itwAvailable.restype = C.c_int
        L.itwSetErrorMode.argtypes = [C.c_int]
        L.itwSetErrorMode.restype = None
        L.itwLastError.restype = C.c_char_p
        L.itwClearError.restype = None
        L.itwSetBc7Path.argtypes = [C.c_int]
        L.itwSetBc7Path.restype = None
        L.itwSetBc7Pilot.argtypes = [C.c_int]
        L.itwSetBc7Pilot.restype = None
        L.itwVersion.restype = C.c_char_p
        L.itwBandForPart.argtypes = [C.c_int32] * 5 + [C.POINTER(C.c_int32)] * 2
        L.itwBandForPart.restype = C.c_int64
        L.itwBandForPartEx.argtypes = [C.c_int32] * 6 + [C.POINTER(C.c_int32)] * 2
        L.itwBandForPartEx.restype = C.c_int64
        if hooks:
            for n in ("itwTestRcp", "itwTestRsqrt", "itwTestF2I"):
                getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
                getattr(L, n).restype = None
            L.itwTestBc7TwoSubsetBounds.argtypes = [C.POINTER(RgbaSurface), C.c_void_p]
            L.itwTestBc7TwoSubsetBounds.restype = None
            L.itwTestBc45IndexTable.argtypes = [C.c_void_p]
            L.itwTestBc45IndexTable.restype = C.c_int
            L.itwMultiGpuTestInjectFailure.argtypes = [C.c_int, C.c_int, C.c_int]
            L.itwMultiGpuTestInjectFailure.restype = None
        # dispatch layer (itw_dispatch.h)
        L.GetProcessorCount.restype = C.c_int
        L.GetBytesPerBlock.argtypes = [C.c_int]
        L.GetBytesPerBlock.restype = C.c_int
        for n in ("CompressImageMT", "CompressImageST"):
            getattr(L, n).argtypes = [C.POINTER(RgbaSurface), C.c_void_p, C.c_void_p, C.c_int]
            getattr(L, n).restype = C.c_bool
        for n in ["CompressImageBC1", "CompressImageBC3"] + ["CompressImageBC7_" + p for p in BC7_PROFILES] \
                + ["CompressImageBC6H_" + p for p in BC6H_PROFILES]:
            getattr(L, n).argtypes = [C.POINTER(RgbaSurface), C.c_void_p]
            getattr(L, n).restype = None
        L.itwCompressImageSliced.argtypes = [C.POINTER(RgbaSurface), C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_bool,
                                             C.c_int64, C.c_void_p, C.c_void_p]
        L.itwCompressImageSliced.restype = C.c_bool
        L.itwCompressImageSlicedEx.argtypes = [C.POINTER(RgbaSurface), C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
        L.itwCompressImageSlicedEx.restype = C.c_bool
        L.itwSetSliceWindow.argtypes = [C.c_int]
        L.itwSetSliceWindow.restype = None
        L.itwSliceWindow.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64]
        L.itwSliceWindow.restype = C.c_int
        L.itwSliceWindowFor.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int64]
        L.itwSliceWindowFor.restype = C.c_int
        L.itwMultiGpuRanks.restype = C.c_int
        L.itwMultiGpuSetInterleave.argtypes = [C.c_int]
        L.itwMultiGpuSetInterleave.restype = None
        L.itwMultiGpuPieces.argtypes = [C.c_int32, C.c_int, C.c_int]
        L.itwMultiGpuPieces.restype = C.c_int
        L.itwMultiGpuTransport.restype = C.c_char_p
        L.itwMultiGpuPeerLinks.restype = C.c_int
        L.itwCompressImageMultiGPU.argtypes = [C.POINTER(RgbaSurface), C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        L.itwCompressImageMultiGPU.restype = C.c_bool
        L.itwCompressImageMultiGPUEx.argtypes = [C.POINTER(RgbaSurface), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(RgbaSurface),
                                                 C.POINTER(MultiGpuStats)]
        L.itwCompressImageMultiGPUEx.restype = C.c_bool
        L.itwCompressImageMultiGPUBands.argtypes = [C.POINTER(RgbaSurface), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(RgbaSurface), C.c_int,
                                                    C.POINTER(MultiGpuStats)]
        L.itwCompressImageMultiGPUBands.restype = C.c_bool
        L.itwPadToMultipleOf4.argtypes = [C.POINTER(RgbaSurface), C.c_int]
        L.itwPadToMultipleOf4.restype = RgbaSurface
        L.itwFreeSurface.argtypes = [C.POINTER(RgbaSurface)]
        L.itwFreeSurface.restype = None
        L.itwPadToMultipleOf4Device.argtypes = [C.POINTER(RgbaSurface), C.c_int, C.c_void_p]
        L.itwPadToMultipleOf4Device.restype = None
        L.itwConvertToRGBA8Device.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.itwConvertToRGBA8Device.restype = C.c_int
        L.itwConvertToRGBA16FDevice.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.itwConvertToRGBA16FDevice.restype = C.c_int
        L.itwDecodeBlocks.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]
        L.itwDecodeBlocks.restype = C.c_int
        # DDS container (itw_dds.h)
        L.itwDdsLevelBytes.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
        L.itwDdsLevelBytes.restype = C.c_size_t
        for n in ("itwDdsHeaderBytes", "itwDdsFileBytes"):
            getattr(L, n).argtypes = [C.POINTER(DdsDesc)]
            getattr(L, n).restype = C.c_size_t
        L.itwDdsWriteHeader.argtypes = [C.POINTER(DdsDesc), C.c_void_p, C.c_size_t]
        L.itwDdsWriteHeader.restype = C.c_size_t
        L.itwDdsReadHeader.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(DdsDesc)]
        L.itwDdsReadHeader.restype = C.c_size_t
        L.itwDdsWriteFile.argtypes = [C.POINTER(DdsDesc), C.POINTER(C.c_void_p), C.c_size_t, C.c_void_p, C.c_size_t]
        L.itwDdsWriteFile.restype = C.c_size_t
    return L


def version():
    return lib().itwVersion().decode()


def source_sha256():
    """SHA-256 over the kernel sources (csrc/*.hip, *.hpp, *.h, Makefile: names and bytes, sorted).  Profiles taken on the GPU box
    carry it (tools/profile_gpu.sh) and bench.py quotes committed counter values only when it equals the tree it runs from."""
    import hashlib
    d = os.path.normpath(os.path.join(_PKG, "..", "csrc"))
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".h")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


ON_ERROR_ABORT, ON_ERROR_RETURN = 0, 1
BC7_PATH = {"auto": 0, "deep": 1, "wide": 2}


def set_bc7_pilot(percent):
    """itwSetBc7Pilot: threshold of the bounded mode order's pilot in percent (0 = reference order for the rest of the surface, 100 = bounded,
    -1 = no pilot, None = back to the library's preset: ITW_BC7_PILOT_THR or 90); same bytes whatever the value."""
    lib().itwSetBc7Pilot(-2 if percent is None else int(percent))


def set_bc7_path(name):
    """itwSetBc7Path: 'auto' | 'deep' | 'wide' (same bytes either way; tests and probes)."""
    lib().itwSetBc7Path(BC7_PATH[name])


def available():
    """itwAvailable(): True if the current HIP device is a gfx950 this library can run on.  Never aborts."""
    return bool(lib().itwAvailable())


def set_error_mode(mode):
    lib().itwSetErrorMode(mode)


def last_error():
    """Message of the last failed ABI call on this host thread, or None."""
    e = lib().itwLastError()
    return e.decode() if e else None


def device_info():
    return lib().itwDeviceInfo().decode()


def bc7_profile(name):
    """GetProfile_<name> (ispc_texcomp.h:67-79) into a zero-initialised struct."""
    if name not in BC7_PROFILES:
        raise KeyError(name)
    s = Bc7Settings()
    getattr(lib(), "GetProfile_" + name)(C.byref(s))
    return s


def bc6h_profile(name):
    """GetProfile_bc6h_<name> (ispc_texcomp.h:81-85)."""
    if name not in BC6H_PROFILES:
        raise KeyError(name)
    s = Bc6hSettings()
    getattr(lib(), "GetProfile_bc6h_" + name)(C.byref(s))
    return s


def band_for_part(width, height, fmt, part, parts):
    """(first_texel_row, texel_rows, output_byte_offset) of `part` among `parts` (itwBandForPart)."""
    y0, n = C.c_int32(), C.c_int32()
    off = lib().itwBandForPartEx(width, height, BYTES_PER_BLOCK[fmt], part, parts, 1 if fmt in KEEPS_PARTIAL_BLOCKS else 0, C.byref(y0), C.byref(n))
    if off < 0:
        raise ValueError((part, parts))
    return y0.value, n.value, off


def _call(fmt, surf, dst_ptr, settings):
    L = lib()
    if fmt == "bc1":
        L.CompressBlocksBC1(C.byref(surf), dst_ptr)
    elif fmt == "bc3":
        L.CompressBlocksBC3(C.byref(surf), dst_ptr)
    elif fmt == "bc4":
        L.CompressBlocksBC4(C.byref(surf), dst_ptr)
    elif fmt == "bc5":
        L.CompressBlocksBC5(C.byref(surf), dst_ptr)
    elif fmt == "bc7":
        st = settings if isinstance(settings, Bc7Settings) else bc7_profile(settings or "slow")
        L.CompressBlocksBC7(C.byref(surf), dst_ptr, C.byref(st))
    elif fmt == "bc6h":
        st = settings if isinstance(settings, Bc6hSettings) else bc6h_profile(settings or "slow")
        L.CompressBlocksBC6H(C.byref(surf), dst_ptr, C.byref(st))
    else:
        raise ValueError(fmt)


def block_count(fmt, width, height):
    """Blocks a width x height surface encodes to: the ISPC formats drop partial blocks, the DirectXTex ones keep them."""
    if fmt in KEEPS_PARTIAL_BLOCKS:
        return ((width + 3) // 4) * ((height + 3) // 4)
    return (width // 4) * (height // 4)


def compress_numpy(fmt, img, settings=None):
    """Host-pointer path (what the Photoshop plugin does): img (H, W, 4) uint8, or uint16 half bits for bc6h.
    Synchronous; returns a uint8 numpy array of packed blocks."""
    import numpy as np
    assert img.ndim == 3 and img.shape[2] == 4 and img.strides[2] == img.itemsize and img.strides[1] == 4 * img.itemsize
    assert img.dtype == (np.uint16 if fmt == "bc6h" else np.uint8)
    h, w = img.shape[:2]
    out = np.empty(block_count(fmt, w, h) * BYTES_PER_BLOCK[fmt], dtype=np.uint8)
    surf = RgbaSurface(img.ctypes.data, w, h, img.strides[0])
    _call(fmt, surf, out.ctypes.data, settings)
    return out


def compress(fmt, img, settings=None, out=None):
    """Device-resident path: img is a CUDA(HIP) torch tensor (H, W, 4) uint8, or int16/uint16/float16 for bc6h;
    rows may be strided.  Launches asynchronously on torch's current stream and returns a uint8 CUDA tensor."""
    import torch
    assert img.is_cuda and img.dim() == 3 and img.shape[2] == 4
    es = img.element_size()
    assert es == (2 if fmt == "bc6h" else 1), "texel type does not match the format"
    assert img.stride(2) == 1 and img.stride(1) == 4
    h, w = img.shape[:2]
    nbytes = block_count(fmt, w, h) * BYTES_PER_BLOCK[fmt]
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
    assert out.is_cuda and out.numel() >= nbytes and out.is_contiguous()
    L = lib()
    with torch.cuda.device(img.device):
        L.itwSetStream(torch.cuda.current_stream(img.device).cuda_stream)
        surf = RgbaSurface(img.data_ptr(), w, h, img.stride(0) * es)
        _call(fmt, surf, out.data_ptr(), settings)
    return out


def bc7_two_subset_bounds(img):
    """Test hook: the bounded BC7 mode order's lower bound of each two-subset shape (itwTestBc7TwoSubsetBounds).
    img: CUDA tensor (H, W, 4) uint8 -> float32 CUDA tensor (blocks, 64), raster block order."""
    import torch
    assert img.is_cuda and img.dim() == 3 and img.shape[2] == 4 and img.element_size() == 1
    assert img.stride(2) == 1 and img.stride(1) == 4
    h, w = img.shape[:2]
    out = torch.empty(((h // 4) * (w // 4), 64), dtype=torch.float32, device=img.device)
    L = test_lib()
    with torch.cuda.device(img.device):
        L.itwSetStream(torch.cuda.current_stream(img.device).cuda_stream)
        surf = RgbaSurface(img.data_ptr(), w, h, img.stride(0))
        L.itwTestBc7TwoSubsetBounds(C.byref(surf), out.data_ptr())
    e = L.itwLastError()
    if e:
        raise RuntimeError(e.decode())
    return out


def image_func(fmt, profile=None, L=None):
    """Address of the CompressImage* trampoline (win32Threads.h:58-80) for a format / profile, as a void* (L: the library instance, default the product)."""
    name = {"bc1": "CompressImageBC1", "bc3": "CompressImageBC3", "bc4": "CompressImageBC4", "bc5": "CompressImageBC5"}.get(fmt) or \
        ("CompressImageBC7_" if fmt == "bc7" else "CompressImageBC6H_") + (profile or "slow")
    return C.cast(getattr(L or lib(), name), C.c_void_p)


def compress_image(fmt, img, profile=None, multithreaded=True, slice_pixels=0, progress=None, out=None, settings=None):
    """The plugin's save path below the pixel conversion (IntelPlugin.cpp:816-884): slice loop -> CompressImageMT/ST ->
    trampoline -> CompressBlocks* (itwCompressImageSliced; with `settings` -- a Bc7Settings / Bc6hSettings -- through
    itwCompressImageSlicedEx).  img: host numpy (H, W, 4) uint8 / uint16 half bits, or a CUDA torch tensor of that shape
    (then `out` is a CUDA uint8 tensor too unless given).  Returns (ok, blocks)."""
    import numpy as np
    h, w = img.shape[:2]
    nbytes = block_count(fmt, w, h) * BYTES_PER_BLOCK[fmt]
    on_device = hasattr(img, "data_ptr")
    if on_device:
        import torch
        if out is None:
            out = torch.zeros(nbytes, dtype=torch.uint8, device=img.device)
        surf = RgbaSurface(img.data_ptr(), w, h, img.stride(0) * img.element_size())
        lib().itwSetStream(torch.cuda.current_stream(img.device).cuda_stream)
    else:
        if out is None:
            out = np.zeros(nbytes, dtype=np.uint8)
        surf = RgbaSurface(img.ctypes.data, w, h, img.strides[0])
    dst = out.data_ptr() if hasattr(out, "data_ptr") else out.ctypes.data
    cb = PROGRESS_FUNC(progress) if progress else None
    pitch = block_count(fmt, w, 4) * BYTES_PER_BLOCK[fmt]
    if settings is not None:
        ok = lib().itwCompressImageSlicedEx(C.byref(surf), dst, pitch, DXGI_FORMAT[fmt], C.cast(C.byref(settings), C.c_void_p), slice_pixels,
                                            C.cast(cb, C.c_void_p) if cb else None, None)
    else:
        ok = lib().itwCompressImageSliced(C.byref(surf), dst, pitch, image_func(fmt, profile),
                                          DXGI_FORMAT[fmt], multithreaded, slice_pixels, C.cast(cb, C.c_void_p) if cb else None, None)
    return bool(ok), out


def multigpu_sub_bands(fmt, width, height, ranks, L=None):
    """The partition a multi-GPU call uses (itw_multigpu.h): [(j, rank, first_texel_row, texel_rows, output_byte_offset)] for the
    K * ranks sub-bands, K = itwMultiGpuPieces(height, ranks); sub-band j belongs to rank j % ranks."""
    k = (L or lib()).itwMultiGpuPieces(height, ranks, 1 if fmt in KEEPS_PARTIAL_BLOCKS else 0)
    out = []
    for j in range(k * ranks):
        y0, rows, off = band_for_part(width, height, fmt, j, k * ranks)
        out.append((j, j % ranks, y0, rows, off))
    return out


def compress_image_multigpu(fmt, img, profile=None, ranks=0, out=None, bands=None, stats=None, L=None, interleave=None):
    """itwCompressImageMultiGPU[Ex]: img is a host numpy array or a CUDA torch tensor (H, W, 4); the block stream comes back in
    the same kind of container (or in `out`, which may be the other kind).  Synchronous.
    bands: optional list of CUDA tensors, K * ranks of them (K = 1..8, stated by the list's length): sub-band j = block rows
    itwBandForPart(j, K * ranks), resident on the device of rank j % ranks (multigpu_sub_bands() cuts a surface the way a call WITHOUT
    resident bands would); no scatter; `img` may then be a (height, width) tuple; goes through itwCompressImageMultiGPUBands, or -- K = 1 --
    through itwCompressImageMultiGPUEx.  interleave: sets K of calls without resident bands for the process first (1 = the reference's
    contiguous bands).  stats: an optional MultiGpuStats to fill (stats.as_dict()).  L: the library instance (default: the product;
    the failure-injection tests pass test_lib(), whose hook arms that instance)."""
    import numpy as np
    L = L or lib()
    if interleave is not None:
        L.itwMultiGpuSetInterleave(int(interleave))      # process-wide (itw_multigpu.h): K sub-bands per rank
    if bands is not None:
        import torch
        h, w = (img if isinstance(img, tuple) else img.shape[:2])
        if not ranks:
            ranks = len(bands)                           # (K = 1: one surface per rank)
        assert len(bands) % ranks == 0, f"{len(bands)} resident surfaces for {ranks} ranks"
        for b in bands:
            assert b.is_cuda and b.dim() == 3 and b.shape[2] == 4 and b.stride(2) == 1 and b.stride(1) == 4 and b.shape[1] == w
            torch.cuda.synchronize(b.device)
        arr = (RgbaSurface * len(bands))(*[RgbaSurface(b.data_ptr(), w, b.shape[0], b.stride(0) * b.element_size()) for b in bands])
        on_gpu, src_ptr, stride = True, None, w * 4 * bands[0].element_size()
        first_dev = bands[0].device
    else:
        h, w = img.shape[:2]
        arr = None
        on_gpu = not isinstance(img, np.ndarray)
        src_ptr, stride = (img.data_ptr(), img.stride(0) * img.element_size()) if on_gpu else (img.ctypes.data, img.strides[0])
        first_dev = img.device if on_gpu else None
    nbytes = block_count(fmt, w, h) * BYTES_PER_BLOCK[fmt]
    if out is None:
        if on_gpu:
            import torch
            out = torch.empty(nbytes, dtype=torch.uint8, device=first_dev)
        else:
            out = np.empty(nbytes, dtype=np.uint8)
    dst_ptr = out.ctypes.data if isinstance(out, np.ndarray) else out.data_ptr()
    if on_gpu and bands is None:
        import torch
        torch.cuda.synchronize(img.device)            # the rank threads read the texels on their own streams
    surf = RgbaSurface(src_ptr, w, h, stride)
    if bands is None and stats is None:
        ok = L.itwCompressImageMultiGPU(C.byref(surf), dst_ptr, image_func(fmt, profile, L), DXGI_FORMAT[fmt], ranks)
    elif bands is not None and len(bands) != ranks:
        ok = L.itwCompressImageMultiGPUBands(C.byref(surf), dst_ptr, image_func(fmt, profile, L), DXGI_FORMAT[fmt], ranks, arr, len(bands),
                                             C.byref(stats) if stats is not None else None)
    else:
        ok = L.itwCompressImageMultiGPUEx(C.byref(surf), dst_ptr, image_func(fmt, profile, L), DXGI_FORMAT[fmt], ranks, arr,
                                          C.byref(stats) if stats is not None else None)
    if not ok:
        e = L.itwLastError()
        raise RuntimeError((e.decode() if e else None) or "itwCompressImageMultiGPU failed")
    return out


def pad_to_multiple_of_4(img):
    """Host pre-pass (IntelPlugin.cpp:893-928) through the library; returns a new numpy array."""
    import numpy as np
    h, w = img.shape[:2]
    ps = 4 * img.itemsize
    surf = RgbaSurface(img.ctypes.data, w, h, img.strides[0])
    out = lib().itwPadToMultipleOf4(C.byref(surf), ps)
    n = out.height * out.stride
    arr = np.ctypeslib.as_array(C.cast(out.ptr, C.POINTER(C.c_uint8)), shape=(n,)).copy()
    lib().itwFreeSurface(C.byref(out))
    return arr.view(img.dtype).reshape(out.height, out.width, 4)


def dds_file(fmt_key, width, height, levels, mip_levels=1, cubemap=False, array_size=1):
    """DDS file bytes for block arrays `levels` (file order).  fmt_key: a key of DXGI_FORMAT."""
    import numpy as np
    d = DdsDesc(width, height, mip_levels, DXGI_FORMAT[fmt_key], 1 if cubemap else 0, array_size)
    total = lib().itwDdsFileBytes(C.byref(d))
    if not total:
        raise ValueError("unsupported DDS description")
    out = np.empty(total, dtype=np.uint8)
    keep = [np.ascontiguousarray(np.asarray(l, dtype=np.uint8)) for l in levels]
    ptrs = (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep])
    n = lib().itwDdsWriteFile(C.byref(d), ptrs, len(keep), out.ctypes.data, out.size)
    if n != total:
        raise ValueError("itwDdsWriteFile failed (level count / sizes)")
    return out


def decode(fmt, blocks, width, height, want_modes=False):
    """GPU decode (itwDecodeBlocks).  blocks: uint8 numpy array or CUDA torch tensor of packed blocks.  Returns texels as
    (H, W, 4) uint8 -- uint16 half bit patterns for bc6h -- in the same kind of container, plus the per-block modes
    (int32) when asked."""
    import numpy as np
    key = {"bc1": 71, "bc3": 77, "bc7": 98, "bc6h": 95, "bc4": 80, "bc5": 83}[fmt]
    nb = block_count(fmt, width, height)
    es = 2 if fmt == "bc6h" else 1
    if isinstance(blocks, np.ndarray):
        blk = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1)
        out = np.empty((height, width, 4), dtype=np.uint16 if fmt == "bc6h" else np.uint8)
        modes = np.empty(nb, dtype=np.int32) if want_modes else None
        rc = lib().itwDecodeBlocks(key, blk.ctypes.data, width, height, out.ctypes.data, width * 4 * es,
                                   modes.ctypes.data if want_modes else None)
    else:
        import torch
        assert blocks.is_cuda and blocks.dtype == torch.uint8 and blocks.is_contiguous()
        out = torch.empty((height, width, 4), dtype=torch.int16 if fmt == "bc6h" else torch.uint8, device=blocks.device)
        modes = torch.empty(nb, dtype=torch.int32, device=blocks.device) if want_modes else None
        with torch.cuda.device(blocks.device):
            lib().itwSetStream(torch.cuda.current_stream(blocks.device).cuda_stream)
            rc = lib().itwDecodeBlocks(key, blocks.data_ptr(), width, height, out.data_ptr(), width * 4 * es,
                                       modes.data_ptr() if want_modes else None)
    if rc != 0:
        raise ValueError("itwDecodeBlocks: unsupported format or size")
    return (out, modes) if want_modes else out
