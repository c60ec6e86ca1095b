"""The plugin's slice loop (IntelPlugin.cpp:851-879) through itwCompressImageSliced at 4096^2 (64 slices of 0x40000 pixels), host
pointers, a progress callback installed: the pipeline (default window, and a sweep of window sizes) next to the literal loop
(itwSetSliceWindow(-1)) and to ONE CompressImageST call over the surface.  Every variant's bytes are compared with the whole-surface
call's.  One JSON object per line.  Usage: python tools/sliced_timing.py [size] [reps] [windows: "0,1,4,16"] [trampolines]"""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np                      # noqa: E402
import itw_amd                          # noqa: E402
from itw_amd import surfaces           # noqa: E402

# the trampolines IntelPlugin.cpp:816-848 selects, + `slow` (BASELINE's profile)
PLUGIN = [("bc1", None), ("bc3", None), ("bc7", "veryfast"), ("bc7", "basic"), ("bc7", "alpha_veryfast"), ("bc7", "alpha_basic"),
          ("bc7", "slow"), ("bc6h", "fast"), ("bc6h", "slow")]


def time_call(fn, reps):
    fn()                                 # warm (first call sizes the staging buffers)
    best = None
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best * 1e3


def _window(L, fmt, prof, fcode, w, h):
    """W of the call (BC7 settings that scan every two-subset shape take larger windows: itwSliceWindowFor)"""
    if fmt == "bc7":
        st = itw_amd.bc7_profile(prof)
        return L.itwSliceWindowFor(fcode, C.cast(C.byref(st), C.c_void_p), w, h, 0)
    return L.itwSliceWindow(fcode, w, h, 0)


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    windows = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
    only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
    L = itw_amd.lib()
    ldr = surfaces.ldr_smooth(size, size)
    if os.environ.get("SLICED_CONTENT") == "baboon":      # the reference's baboon.png tiled (opaque): the natural-image end of BC7 `slow`
        z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
        reps = -(-size // z["baboon"].shape[0])
        ldr = np.ascontiguousarray(np.tile(z["baboon"], (reps, reps, 1))[:size, :size])
    rgba = ldr                          # the generator's alpha channel is a second field (translucent)
    hdr = surfaces.hdr_smooth(size, size)
    for fmt, prof in PLUGIN:
        name = fmt + ("_" + prof if prof else "")
        if only and name not in only:
            continue
        img = hdr if fmt == "bc6h" else (rgba if prof and prof.startswith("alpha") else ldr)
        h, w = img.shape[:2]
        nbytes = itw_amd.block_count(fmt, w, h) * itw_amd.BYTES_PER_BLOCK[fmt]
        out = np.zeros(nbytes, dtype=np.uint8)
        want = np.zeros(nbytes, dtype=np.uint8)
        surf = itw_amd.RgbaSurface(img.ctypes.data, w, h, img.strides[0])
        fn = itw_amd.image_func(fmt, prof)
        fcode = itw_amd.DXGI_FORMAT[fmt]
        pitch = itw_amd.block_count(fmt, w, 4) * itw_amd.BYTES_PER_BLOCK[fmt]
        mpix = w * h / 1e6
        ms = time_call(lambda: L.CompressImageST(C.byref(surf), want.ctypes.data, fn, fcode), reps)
        print(json.dumps({"trampoline": name, "shape": "one call", "ms": round(ms, 3), "mpix_s": round(mpix / ms * 1e3, 1)}), flush=True)
        calls = []
        cb = itw_amd.PROGRESS_FUNC(lambda i, n, u: calls.append(i) or True)
        for W in [-1] + windows:
            L.itwSetSliceWindow(W)
            out[:] = 0
            del calls[:]
            run = lambda: L.itwCompressImageSliced(C.byref(surf), out.ctypes.data, pitch, fn, fcode, False, 0, C.cast(cb, C.c_void_p), None)
            ms = time_call(run, reps)
            slices = max(1, w * h // 0x40000)
            per_run = len(calls) // (reps + 1)
            print(json.dumps({"trampoline": name, "shape": "literal loop" if W < 0 else "pipeline", "window": _window(L, fmt, prof, fcode, w, h),
                              "slices": slices, "progress_calls": per_run, "ms": round(ms, 3), "mpix_s": round(mpix / ms * 1e3, 1),
                              "bytes_equal_one_call": bool(np.array_equal(out, want))}), flush=True)
        L.itwSetSliceWindow(0)


if __name__ == "__main__":
    main()
