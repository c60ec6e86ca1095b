"""Throughput of the host-pointer path at the call granularities of the reference's callers (4096^2, pageable memory):
  whole  : one CompressBlocks* call for the surface (what INTEGRATION.md recommends)
  slices : the plugin's 0x40000-pixel slice loop (IntelPlugin.cpp:851) through this library's dispatch layer, which
           encodes each slice in one call per GPU
  legacy : the same slices cut again into ITW_WORKERS bands, one host thread each (win32Threads.cpp:217 on a 64-thread host)
Run with ITW_WORKERS=64 to reproduce the legacy column."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np
import itw_amd
from itw_amd import surfaces

size = 4096
workers = itw_amd.lib().GetProcessorCount()
for fmt, prof in (("bc1", None), ("bc7", "basic"), ("bc7", "slow"), ("bc6h", "slow")):
    img = surfaces.hdr_smooth(size, size) if fmt == "bc6h" else surfaces.ldr_smooth(size, size)
    def t(fn, n=3):
        fn(); best = 1e9
        for _ in range(n):
            t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
        return best
    whole = t(lambda: itw_amd.compress_numpy(fmt, img, prof))
    sliced = t(lambda: itw_amd.compress_image(fmt, img, prof, multithreaded=(workers > 1), slice_pixels=0), n=2)
    big = t(lambda: itw_amd.compress_image(fmt, img, prof, multithreaded=(workers > 1), slice_pixels=1 << 26), n=2)
    print(f"workers={workers:3d} {fmt:5s} {prof or '-':6s} whole call {whole*1e3:8.2f} ms | 0x40000-px slices {sliced*1e3:9.2f} ms = {size*size/sliced/1e6:8.1f} Mpix/s"
          f" | one 64 Mpix slice {big*1e3:8.2f} ms", flush=True)
itw_amd.lib().DestroyThreads()
