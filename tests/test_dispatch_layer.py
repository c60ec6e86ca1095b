"""The dispatch layer above the ABI (include/itw_dispatch.h = win32Threads.h:24-80 restated portably): band rule,
bytes per block, host pad pre-pass (CPU); on the GPU: every trampoline and CompressImageMT/ST byte-equal to the direct
CompressBlocks* call and to the oracle, the slice loop's progress / early-out contract, and the device pad kernel."""
import ctypes as C

import numpy as np
import pytest


def test_bytes_per_block_and_worker_count(itw):
    L = itw.lib()
    assert [L.GetBytesPerBlock(f) for f in (71, 72, 77, 78, 95, 96, 98, 99, 0, 28)] == [8, 8, 16, 16, 16, 16, 16, 16, 8, 8]
    assert [L.GetBytesPerBlock(f) for f in (80, 83)] == [8, 16]      # BC4 (the reference's default), BC5 (extension)
    assert L.GetProcessorCount() >= 1                        # "1 or more" (win32Threads.h:51)


@pytest.mark.parametrize("shape,dtype", [((7, 5), np.uint8), ((4, 4), np.uint8), ((1, 1), np.uint16), ((10, 13), np.uint16),
                                         ((339, 127), np.uint8)])
def test_host_pad_replicates_edges(itw, shape, dtype):
    """IntelPlugin.cpp:893-928: new texels copy the last texel of the row, new rows copy the last row."""
    rng = np.random.default_rng(5)
    h, w = shape
    img = rng.integers(0, np.iinfo(dtype).max, size=(h, w, 4), dtype=dtype)
    got = itw.pad_to_multiple_of_4(img)
    H, W = (h + 3) & ~3, (w + 3) & ~3
    want = np.pad(img, ((0, H - h), (0, W - w), (0, 0)), mode="edge")
    assert got.shape == (H, W, 4) and np.array_equal(got, want)


def test_host_pad_respects_stride(itw):
    rng = np.random.default_rng(6)
    big = rng.integers(0, 255, size=(9, 16, 4), dtype=np.uint8)
    view = big[:, 2:9]                                        # 7 wide, row stride of 16 texels
    assert np.array_equal(itw.pad_to_multiple_of_4(view), np.pad(view, ((0, 3), (0, 1), (0, 0)), mode="edge"))


def test_slice_window_rule(itw):
    """itwSliceWindow (no GPU needed): W = slices per window of the pipelined slice loop -- about 131 072 blocks for BC7 / BC6H, 262 144 for the
    PCIe-bound formats and the heavy BC7 settings; itwSetSliceWindow fixes it, -1 turns the pipeline off (0)."""
    import os
    if "ITW_SLICE_WINDOW" in os.environ or os.environ.get("ITW_SLICED_PIPELINE") == "0":
        pytest.skip("the environment presets the window")
    L = itw.lib()
    try:
        assert [L.itwSliceWindow(f, 4096, 4096, 0) for f in (98, 99, 95, 96)] == [8, 8, 8, 8]          # 64 slices of 16 384 blocks -> 131 072 per window
        assert [L.itwSliceWindow(f, 4096, 4096, 0) for f in (71, 77, 80, 83)] == [16, 16, 16, 16]      # PCIe-bound: 262 144
        assert L.itwSliceWindow(98, 16384, 16384, 0) == 8 and L.itwSliceWindow(71, 16384, 16384, 0) == 16
        assert L.itwSliceWindow(98, 2048, 2048, 0) == 8 and L.itwSliceWindow(98, 1024, 1024, 0) == 4     # 16 / 4 slices: two windows / one (never smaller ones)
        assert L.itwSliceWindow(98, 4096, 4096, 1 << 20) == 2                                            # 16 slices of 65 536 blocks
        assert L.itwSliceWindow(98, 64, 64, 0) == 1                                                      # one slice
        # BC7 settings whose modes 1/3 scan every two-subset shape (twice the work per block): windows twice as large
        for name, want in (("slow", 16), ("alpha_slow", 16), ("basic", 8), ("veryfast", 8), ("alpha_basic", 8)):
            st = itw.bc7_profile(name)
            assert L.itwSliceWindowFor(98, C.cast(C.byref(st), C.c_void_p), 4096, 4096, 0) == want, name
        assert L.itwSliceWindowFor(71, None, 4096, 4096, 0) == L.itwSliceWindow(71, 4096, 4096, 0) == 16
        L.itwSetSliceWindow(5)
        assert L.itwSliceWindow(98, 4096, 4096, 0) == 5 and L.itwSliceWindow(98, 1024, 1024, 0) == 4     # clamped to the slice count
        L.itwSetSliceWindow(-1)
        assert L.itwSliceWindow(98, 4096, 4096, 0) == 0
    finally:
        L.itwSetSliceWindow(0)


# ---- GPU ------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_trampolines_equal_direct_calls_and_oracle(itw, gpu, oracle):
    from itw_amd import surfaces
    ldr = surfaces.ldr_smooth(64, 96)
    hdr = surfaces.hdr_smooth(32, 64)
    odd = np.ascontiguousarray(ldr[:61, :90])                 # the DirectXTex formats keep partial blocks
    cases = [("bc1", ldr, None), ("bc3", ldr, None)] + [("bc7", ldr, p) for p in itw.BC7_PROFILES] \
        + [("bc6h", hdr, p) for p in itw.BC6H_PROFILES] + [("bc4", ldr, None), ("bc5", ldr, None), ("bc4", odd, None), ("bc5", odd, None)]
    for fmt, img, prof in cases:
        h, w = img.shape[:2]
        out = np.zeros(itw.block_count(fmt, w, h) * itw.BYTES_PER_BLOCK[fmt], dtype=np.uint8)
        surf = itw.RgbaSurface(img.ctypes.data, w, h, img.strides[0])
        fn = itw.image_func(fmt, prof)
        for entry in ("CompressImageMT", "CompressImageST"):
            out[:] = 0
            assert getattr(itw.lib(), entry)(C.byref(surf), out.ctypes.data, fn, itw.DXGI_FORMAT[fmt])
            assert np.array_equal(out, itw.compress_numpy(fmt, img, prof)), (entry, fmt, prof)
        assert np.array_equal(out, oracle.encode(fmt, img, prof).reshape(-1)), (fmt, prof)


@pytest.fixture
def slice_window(itw):
    """itwSetSliceWindow for one test, back to the default afterwards."""
    yield itw.lib().itwSetSliceWindow
    itw.lib().itwSetSliceWindow(0)


@pytest.mark.gpu
def test_slice_loop_progress_and_abort(itw, gpu, oracle, slice_window):
    """IntelPlugin.cpp:851-879: `slices = w*h / slice_pixels`, progress polled before every slice but the first, a false
    return stops the job and leaves the slices already written -- with a window of one slice exactly those (the default window of so small
    a job is one slice too; the environment matrix presets others, hence the explicit 1)."""
    from itw_amd import surfaces
    slice_window(1)
    img = surfaces.ldr_smooth(128, 64)                        # 8192 px
    want = oracle.encode("bc7", img, "veryfast").reshape(-1)
    calls = []
    ok, out = itw.compress_image("bc7", img, "veryfast", slice_pixels=2048, progress=lambda i, n, u: calls.append((i, n)) or True)
    assert ok and calls == [(1, 4), (2, 4), (3, 4)] and np.array_equal(out, want)
    ok, out = itw.compress_image("bc7", img, "veryfast", multithreaded=False, slice_pixels=2048, progress=lambda i, n, u: i < 2)
    assert not ok
    rows = 128 // 4                                           # 32 block rows, 8 per slice, 16 blocks of 16 B per row
    done = 2 * 8 * 16 * 16
    assert np.array_equal(out[:done], want[:done]) and not out[done:].any()
    ok, out = itw.compress_image("bc1", img)                  # default slice size: one slice, no callback needed
    assert ok and np.array_equal(out, oracle.encode("bc1", img).reshape(-1))
    odd = np.ascontiguousarray(img[:126, :61])                # BC5, partial last block row / column, 3 slices
    ok, out = itw.compress_image("bc5", odd, slice_pixels=2048, progress=lambda i, n, u: True)
    assert ok and np.array_equal(out, oracle.encode("bc5", odd).reshape(-1))


PLUGIN_TRAMPOLINES = [("bc1", None), ("bc3", None), ("bc7", "veryfast"), ("bc7", "basic"), ("bc7", "alpha_veryfast"), ("bc7", "alpha_basic"),
                      ("bc6h", "fast"), ("bc6h", "slow"), ("bc7", "slow"), ("bc7", "alpha_slow"), ("bc4", None), ("bc5", None)]


def _default_window_rule():
    """False when the environment presets the window (tools/gpu_env_matrix.sh runs this file under ITW_SLICE_WINDOW / ITW_SLICED_PIPELINE):
    the asserts on the DEFAULT rule's values are then skipped; bytes and progress calls must hold either way."""
    import os
    return "ITW_SLICE_WINDOW" not in os.environ and os.environ.get("ITW_SLICED_PIPELINE", "1") != "0"


@pytest.mark.gpu
@pytest.mark.parametrize("resident", [False, True])
def test_sliced_pipeline_every_plugin_trampoline_vs_oracle(itw, gpu, oracle, resident, slice_window):
    """itwCompressImageSliced as a pipeline (windows of slices in flight, itw_dispatch.h): every trampoline the plugin selects
    (IntelPlugin.cpp:816-848) + the slow presets + BC4/BC5 on 64 slices, host and device pointers: bytes of the oracle, one progress
    call per slice boundary, in order."""
    import torch
    from itw_amd import surfaces
    ldr = surfaces.ldr_smooth(384, 256)                       # 98304 px -> 64 slices of 1536 px = 6 texel rows: slices of 4 and 8 rows alternate
    hdr = surfaces.hdr_smooth(384, 256)
    odd = np.ascontiguousarray(ldr[:382, :253])               # partial last block row / column (BC4 / BC5)
    slice_window(8)                                           # (the default rule would make ONE window of so small a surface)
    for fmt, prof in PLUGIN_TRAMPOLINES:
        img = hdr if fmt == "bc6h" else (odd if fmt in ("bc4", "bc5") else ldr)
        h, w = img.shape[:2]
        slice_pixels = w * h // 64
        want = oracle.encode(fmt, img, prof).reshape(-1)
        calls = []
        src = torch.from_numpy(img.view(np.int16) if fmt == "bc6h" else img).to(gpu) if resident else img
        ok, out = itw.compress_image(fmt, src, prof, multithreaded=False, slice_pixels=slice_pixels, progress=lambda i, n, u: calls.append((i, n)) or True)
        got = out.cpu().numpy() if resident else out
        assert ok and calls == [(i, 64) for i in range(1, 64)], (fmt, prof, calls[:5])
        assert np.array_equal(got, want), (fmt, prof, resident)
    # every window size gives the same bytes (0 = the default rule: one window here; 1 = the reference's granularity)
    want = oracle.encode("bc7", ldr, "basic").reshape(-1)
    for W in (0, 1, 3, 5, 64):
        slice_window(W)
        src = torch.from_numpy(ldr).to(gpu) if resident else ldr
        ok, out = itw.compress_image("bc7", src, "basic", slice_pixels=1536, progress=lambda i, n, u: True)
        assert ok and np.array_equal(out.cpu().numpy() if resident else out, want), W


@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 32, 63])
def test_sliced_pipeline_abort_contract(itw, gpu, oracle, k, slice_window):
    """progress(k) returns false = "abort after slice k-1": the call returns false after exactly k progress calls, slices < k hold the
    oracle's bytes, at most the rest of slice k-1's window is written behind them and nothing of the window in flight is copied back
    (host memory); with a window of one slice it is the reference's early out to the byte (IntelPlugin.cpp:857-858)."""
    import torch
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(512, 256)                       # 64 slices of 8 texel rows = 2 block rows of 64 blocks
    want = oracle.encode("bc7", img, "veryfast").reshape(-1)
    slice_bytes = 2 * 64 * 16
    for W in (8, 1):                                          # windows of 8 slices (what a 4096^2 save gets) and the reference's granularity
        slice_window(W)
        win = itw.lib().itwSliceWindow(98, 256, 512, 2048)
        assert win in (W, 0)
        if win == 0:                                          # ITW_SLICED_PIPELINE=0 (environment matrix): the literal loop -- the reference's granularity
            win = 1
        for resident in (False, True):
            calls = []
            src = torch.from_numpy(img).to(gpu) if resident else img
            ok, out = itw.compress_image("bc7", src, "veryfast", multithreaded=False, slice_pixels=2048,
                                         progress=lambda i, n, u: calls.append(i) or i != k)
            got = out.cpu().numpy() if resident else out
            assert not ok and calls == list(range(1, k + 1)), (W, resident, calls)
            assert itw.lib().itwLastError() is None            # an early out is not a failure
            done = k * slice_bytes
            assert np.array_equal(got[:done], want[:done]), (W, resident)
            window_end = ((k - 1) // win + 1) * win * slice_bytes   # end of the window slice k-1 belongs to
            assert np.array_equal(got[done:window_end], want[done:window_end])
            if not resident:                                  # the window in flight is drained, not copied back
                assert not got[window_end:].any(), (W, k)
    # the next call on the same thread starts clean (streams drained, workspace ordered)
    slice_window(0)
    ok, out = itw.compress_image("bc7", img, "veryfast", slice_pixels=2048, progress=lambda i, n, u: True)
    assert ok and np.array_equal(out, want)


@pytest.mark.gpu
def test_sliced_ex_with_a_settings_struct_and_literal_loop_switch(itw, gpu, oracle, slice_window):
    """itwCompressImageSlicedEx takes the caller's settings struct; itwSetSliceWindow(-1) turns itwCompressImageSliced back into the
    literal loop (same bytes, same progress calls); a caller's own CompressionFunc is opaque and always takes the literal loop."""
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(256, 256)
    st = itw.bc7_profile("basic")
    st.fastSkipTreshold_mode1 = 5; st.fastSkipTreshold_mode3 = 3; st.refineIterations[1] = 1
    want = oracle.encode("bc7", img, _as_oracle_settings(oracle, st)).reshape(-1)
    calls = []
    ok, out = itw.compress_image("bc7", img, slice_pixels=4096, settings=st, progress=lambda i, n, u: calls.append(i) or True)
    assert ok and calls == list(range(1, 16)) and np.array_equal(out, want)
    hdr = surfaces.hdr_smooth(128, 256)
    s6 = itw.bc6h_profile("basic"); s6.fastSkipTreshold = 3
    ok, out = itw.compress_image("bc6h", hdr, slice_pixels=2048, settings=s6, progress=lambda i, n, u: True)
    assert ok and np.array_equal(out, oracle.encode("bc6h", hdr, _as_oracle_settings(oracle, s6)).reshape(-1))
    slice_window(-1)
    assert itw.lib().itwSliceWindow(98, 256, 256, 4096) == 0
    calls = []
    want = oracle.encode("bc7", img, "alpha_basic").reshape(-1)
    ok, out = itw.compress_image("bc7", img, "alpha_basic", slice_pixels=4096, progress=lambda i, n, u: calls.append(i) or True)
    assert ok and calls == list(range(1, 16)) and np.array_equal(out, want)
    slice_window(0)
    # an opaque CompressionFunc (here a ctypes callback that forwards to the library): literal loop, one call per slice
    seen = []
    def own(surf_p, dst):
        seen.append(surf_p.contents.height)
        itw.lib().CompressImageBC7_alpha_basic(surf_p, dst)
    fn = C.CFUNCTYPE(None, C.POINTER(itw.RgbaSurface), C.c_void_p)(own)
    out = np.zeros_like(want)
    surf = itw.RgbaSurface(img.ctypes.data, 256, 256, img.strides[0])
    assert itw.lib().itwCompressImageSliced(C.byref(surf), out.ctypes.data, 64 * 16, C.cast(fn, C.c_void_p), 98, False, 4096, None, None)
    assert seen == [16] * 16 and np.array_equal(out, want)


def _as_oracle_settings(oracle, st):
    """The binding's settings struct as the oracle's (same layout, ispc_texcomp.h:27-50)."""
    cls = oracle.Bc7Settings if C.sizeof(st) == 64 else oracle.Bc6hSettings
    o = cls()
    C.memmove(C.byref(o), C.byref(st), C.sizeof(st))
    return o


@pytest.mark.gpu
def test_device_pad_kernel(itw, gpu):
    import torch
    rng = np.random.default_rng(8)
    for (h, w), dtype in (((7, 5), np.uint8), ((339, 127), np.uint8), ((10, 13), np.int16), ((4, 8), np.uint8)):
        img = rng.integers(0, 127, size=(h, w, 4)).astype(dtype)
        d_in = torch.from_numpy(img).to(gpu)
        H, W = (h + 3) & ~3, (w + 3) & ~3
        d_out = torch.zeros((H, W, 4), dtype=d_in.dtype, device=gpu)
        itw.lib().itwSetStream(torch.cuda.current_stream().cuda_stream)
        surf = itw.RgbaSurface(d_in.data_ptr(), w, h, d_in.stride(0) * d_in.element_size())
        itw.lib().itwPadToMultipleOf4Device(C.byref(surf), 4 * d_in.element_size(), d_out.data_ptr())
        torch.cuda.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), np.pad(img, ((0, H - h), (0, W - w), (0, 0)), mode="edge"))


@pytest.mark.gpu
def test_pad_then_encode_device_resident(itw, gpu, oracle):
    """The pre-pass feeding the ABI without leaving HBM: a 127 x 339 surface (landscape-detail.jpg's size)."""
    import torch
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(340, 128)[:339, :127].copy()
    d_in = torch.from_numpy(img).to(gpu)
    d_pad = torch.empty((340, 128, 4), dtype=torch.uint8, device=gpu)
    itw.lib().itwSetStream(torch.cuda.current_stream().cuda_stream)
    surf = itw.RgbaSurface(d_in.data_ptr(), 127, 339, 127 * 4)
    itw.lib().itwPadToMultipleOf4Device(C.byref(surf), 4, d_pad.data_ptr())
    got = itw.compress("bc3", d_pad).cpu().numpy()
    want = oracle.encode("bc3", itw.pad_to_multiple_of_4(img)).reshape(-1)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_eight_workers_on_one_device(gpu, oracle):
    """The 8-band path of CompressImageMT (win32Threads.cpp:217-231) on a one-GPU box: ITW_WORKERS=8 starts eight host
    workers that share device 0; unequal and empty bands included (heights that do not divide by 32)."""
    import os
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import itw_amd
from itw_amd import surfaces
from oracle import pyoracle
L = itw_amd.lib()
assert L.GetProcessorCount() == 8
for h, w, fmt, prof in ((100, 64, "bc1", None), (36, 32, "bc3", None), (256, 64, "bc7", "veryfast"), (8, 16, "bc7", "alpha_basic"), (64, 32, "bc6h", "fast"),
                        (102, 63, "bc4", None), (35, 30, "bc5", None), (3, 5, "bc5", None)):
    H, W = (h + 3) // 4 * 4, (w + 3) // 4 * 4
    img = surfaces.hdr_smooth(h, w) if fmt == "bc6h" else np.ascontiguousarray(surfaces.ldr_smooth(H, W)[:h, :w])
    out = np.zeros(itw_amd.block_count(fmt, w, h) * itw_amd.BYTES_PER_BLOCK[fmt], dtype=np.uint8)
    surf = itw_amd.RgbaSurface(img.ctypes.data, w, h, img.strides[0])
    for rep in range(3):
        out[:] = 0
        assert L.CompressImageMT(C.byref(surf), out.ctypes.data, itw_amd.image_func(fmt, prof), itw_amd.DXGI_FORMAT[fmt])
        assert np.array_equal(out, pyoracle.encode(fmt, img, prof).reshape(-1)), (h, w, fmt, prof, rep)
L.DestroyThreads()
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ITW_WORKERS="8")
    r = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "intel-texture-works-plugin_amd"))],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_legacy_call_granularity_64_workers(gpu, oracle):
    """What the unmodified plugin does on a 64-thread host (hard part 4): 0x40000-pixel slices
    (IntelPlugin.cpp:851) each cut into 64 bands (win32Threads.cpp:217) -> host-pointer calls of ~4 k pixels from 64
    threads at once, all landing on one device.  Streams, staging buffers and the BC7 workspace are per host thread."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import itw_amd
from itw_amd import surfaces
from oracle import pyoracle
assert itw_amd.lib().GetProcessorCount() == 64
img = surfaces.ldr_smooth(1024, 1024)                      # 4 slices x 64 bands of 4 rows
for fmt, prof in (("bc1", None), ("bc7", "veryfast"), ("bc7", "alpha_basic"), ("bc5", None)):
    calls = []
    ok, out = itw_amd.compress_image(fmt, img, prof, multithreaded=True, slice_pixels=0, progress=lambda i, n, u: calls.append(n) or True)
    assert ok and calls == [4, 4, 4]
    assert np.array_equal(out, pyoracle.encode_mt(fmt, img, prof).reshape(-1)), (fmt, prof)
hdr = surfaces.hdr_smooth(512, 512)
ok, out = itw_amd.compress_image("bc6h", hdr, "basic")
assert ok and np.array_equal(out, pyoracle.encode_mt("bc6h", hdr, "basic").reshape(-1))
itw_amd.lib().DestroyThreads()
print("ok")
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ITW_WORKERS="64", ITW_SLICED_PIPELINE="0")     # the literal loop: one CompressImageMT per slice, 64 bands each
    r = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "intel-texture-works-plugin_amd"))],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("workers", [2, 3, 8])
def test_sliced_pipeline_one_pipeline_per_worker(gpu, oracle, workers):
    """Several GPUs behind itwCompressImageSliced (here: ITW_WORKERS workers sharing device 0): the windows are dealt round-robin to the
    pool's workers, each runs its share as its own pipeline, and the submitting thread calls progress in the reference's order.  Bytes of the
    oracle; one call per slice boundary, ascending; an abort stops every worker, returns false and leaves slices < k written."""
    import os
    import subprocess
    import sys
    code = r"""
import ctypes as C, sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
import itw_amd
from itw_amd import surfaces
from oracle import pyoracle
L = itw_amd.lib()
n = L.GetProcessorCount()
L.itwSetSliceWindow(4)                                    # 16 windows for the workers to share (the default rule would make one of so small a surface)
ldr = surfaces.ldr_smooth(512, 256)                       # 64 slices of 2048 px (8 texel rows)
hdr = surfaces.hdr_smooth(256, 128)
for fmt, img, prof, px in (("bc7", ldr, "basic", 2048), ("bc7", ldr, "alpha_basic", 2048), ("bc1", ldr, None, 2048), ("bc6h", hdr, "slow", 512),
                           ("bc5", np.ascontiguousarray(ldr[:509, :253]), None, 2012)):
    h, w = img.shape[:2]
    assert w * h // px == 64
    want = pyoracle.encode(fmt, img, prof).reshape(-1)
    calls = []
    ok, out = itw_amd.compress_image(fmt, img, prof, multithreaded=True, slice_pixels=px, progress=lambda i, t, u: calls.append(i) or True)
    assert ok and calls == list(range(1, 64)), (fmt, prof, calls[:8])
    assert np.array_equal(out, want), (fmt, prof)
for k in (1, 9, 40, 63):
    want = pyoracle.encode("bc7", ldr, "veryfast").reshape(-1)
    calls = []
    ok, out = itw_amd.compress_image("bc7", ldr, "veryfast", multithreaded=True, slice_pixels=2048, progress=lambda i, t, u: calls.append(i) or i != k)
    done = k * 2 * 64 * 16
    assert not ok and calls == list(range(1, k + 1)) and np.array_equal(out[:done], want[:done]), (k, calls[-3:])
    assert itw_amd.lib().itwLastError() is None
    written = out.reshape(-1, 2048).any(axis=1)            # per slice: anything written?
    assert all(np.array_equal(out[s * 2048:(s + 1) * 2048], want[s * 2048:(s + 1) * 2048]) for s in np.nonzero(written)[0])   # whatever arrived is right
ok, out = itw_amd.compress_image("bc7", ldr, "veryfast", multithreaded=True, slice_pixels=2048)      # and the pool is fine afterwards
assert ok and np.array_equal(out, pyoracle.encode("bc7", ldr, "veryfast").reshape(-1))
L.DestroyThreads()
print("ok", n)
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, ITW_WORKERS=str(workers))
    env.pop("ITW_SLICED_PIPELINE", None); env.pop("ITW_SLICE_WINDOW", None)
    r = subprocess.run([sys.executable, "-c", code % (root, os.path.join(root, "intel-texture-works-plugin_amd"))],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and ("ok %d" % workers) in r.stdout, r.stdout + r.stderr
