"""GPU parity, BC1/BC3: the HIP kernels (csrc/bc1_bc3.hip, called through the C ABI) emit byte-identical
blocks to the oracle (oracle/bc1_bc3.c, restating kernel.ispc:231-614) and to the committed golden streams.
Bar: bit-exact (integer output)."""
import numpy as np
import pytest

from conftest import first_mismatch

pytestmark = pytest.mark.gpu
BPB = {"bc1": 8, "bc3": 16}


def gpu_encode(itw, gpu, fmt, img):
    import torch
    t = torch.from_numpy(img).to(gpu)
    out = itw.compress(fmt, t)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("fmt", ["bc1", "bc3"])
@pytest.mark.parametrize("name", ["baboon", "monkey", "edge_cases"])
def test_golden_streams(itw, gpu, golden_inputs, golden_blocks, fmt, name):
    got = gpu_encode(itw, gpu, fmt, golden_inputs[name])
    want = golden_blocks[f"{name}.{fmt}"]
    assert got.size == want.size
    assert first_mismatch(got, want, BPB[fmt]) is None, first_mismatch(got, want, BPB[fmt])


@pytest.mark.parametrize("fmt", ["bc1", "bc3"])
@pytest.mark.parametrize("gen,h,w", [("ldr_smooth", 512, 512), ("ldr_uniform", 256, 512), ("ldr_smooth", 52, 100),
                                     ("ldr_uniform", 4, 4), ("ldr_uniform", 8, 260), ("ldr_smooth", 1024, 1028)])
def test_synthetic_vs_oracle(itw, gpu, oracle, fmt, gen, h, w):
    from itw_amd import surfaces
    img = getattr(surfaces, gen)(h, w)
    got = gpu_encode(itw, gpu, fmt, img)
    want = oracle.encode(fmt, img)
    assert first_mismatch(got, want, BPB[fmt]) is None, first_mismatch(got, want, BPB[fmt])


@pytest.mark.parametrize("fmt", ["bc1", "bc3"])
def test_full_size_4096(itw, gpu, oracle, fmt):
    """BASELINE configs[1]: synthetic 4096x4096 RGBA8.  The oracle finishes a full 4k BC1/BC3 surface in seconds
    (threaded bands), so the whole surface is compared, not a sample."""
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(4096, 4096)
    got = gpu_encode(itw, gpu, fmt, img)
    want = oracle.encode_mt(fmt, img)
    assert first_mismatch(got, want, BPB[fmt]) is None, first_mismatch(got, want, BPB[fmt])


@pytest.mark.parametrize("fmt", ["bc1", "bc3"])
def test_strided_and_unaligned_surfaces(itw, gpu, oracle, fmt):
    """rgba_surface.stride is free (rowPitch, IntelPlugin.cpp:232-240): padded rows, and a base pointer that is
    only 4-byte aligned (forces the non-vector load path)."""
    import torch
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(64, 72)
    want = oracle.encode(fmt, img)
    big = torch.zeros((64, 100, 4), dtype=torch.uint8, device=gpu)
    big[:, 3:75] = torch.from_numpy(img).to(gpu)
    view = big[:, 3:75]                      # stride 400 bytes, base offset 12 bytes
    got = itw.compress(fmt, view)
    torch.cuda.synchronize()
    assert first_mismatch(got.cpu().numpy(), want, BPB[fmt]) is None


@pytest.mark.parametrize("fmt", ["bc1", "bc3"])
def test_host_pointer_path(itw, gpu, oracle, fmt):
    """What the plugin does: pageable host memory in, host memory out, synchronous (ispc_texcomp.cpp:417-425)."""
    from itw_amd import surfaces
    img = surfaces.ldr_uniform(128, 132)
    got = itw.compress_numpy(fmt, img)
    want = oracle.encode(fmt, img)
    assert first_mismatch(got, want, BPB[fmt]) is None
    # row band of a larger surface, like win32Threads.cpp:228-230 hands over
    y0, n, off = itw.band_for_part(132, 128, fmt, 1, 3)
    band = img[y0:y0 + n]
    got_b = itw.compress_numpy(fmt, band)
    assert (got_b == want[off:off + got_b.size]).all()


def test_partial_blocks_are_dropped(itw, gpu, oracle):
    """height not a multiple of 4: the last partial block row is dropped (kernel.ispc:600).  A width that is not a
    multiple of 4 is outside the reference contract (its output pitch becomes width*data_size bytes, kernel.ispc:157,
    i.e. misaligned fractional blocks); the product packs width/4 blocks per row tightly there and is only checked
    for shape."""
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(30, 44)
    got = itw.compress_numpy("bc1", img)
    assert got.size == (30 // 4) * (44 // 4) * 8
    want = oracle.encode("bc1", img)
    assert (got == want).all()
    odd = surfaces.ldr_smooth(30, 45)
    got = itw.compress_numpy("bc1", odd)
    assert got.size == (30 // 4) * (45 // 4) * 8
    assert (got == oracle.encode("bc1", np.ascontiguousarray(odd[:, :44]))).all()


def test_zero_blocks_is_a_no_op(itw, gpu):
    import torch
    t = torch.zeros((3, 3, 4), dtype=torch.uint8, device=gpu)
    out = itw.compress("bc1", t)
    assert out.numel() == 0


def test_concurrent_host_threads(itw, gpu, oracle):
    """The legacy caller invokes the ABI from many threads on disjoint bands of one image at once
    (win32Threads.cpp:211-274)."""
    from concurrent.futures import ThreadPoolExecutor
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(256, 256)
    want = oracle.encode("bc3", img)
    parts = 8

    def work(p):
        y0, n, off = itw.band_for_part(256, 256, "bc3", p, parts)
        return off, itw.compress_numpy("bc3", img[y0:y0 + n])

    with ThreadPoolExecutor(max_workers=parts) as ex:
        res = list(ex.map(work, range(parts)))
    got = np.zeros_like(want)
    for off, blk in res:
        got[off:off + blk.size] = blk
    assert (got == want).all()


def test_bottom_up_and_repeated_rows(itw, gpu, oracle):
    """`stride` is a signed int the reference simply multiplies by y (kernel.ispc:105-151): a negative stride walks a
    bottom-up image, stride 0 repeats one row.  Host pointers (staged row by row) and device pointers alike."""
    import ctypes as C
    import torch
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(64, 96)
    flipped = np.ascontiguousarray(img[::-1])
    want = oracle.encode("bc3", flipped)
    out = np.zeros(want.size, dtype=np.uint8)
    last_row = img.ctypes.data + 63 * img.strides[0]
    surf = itw.RgbaSurface(last_row, 96, 64, -img.strides[0])
    itw.lib().CompressBlocksBC3(C.byref(surf), out.ctypes.data)
    assert first_mismatch(out, want, 16) is None, first_mismatch(out, want, 16)
    d = torch.from_numpy(img).to(gpu)
    d_out = torch.zeros(want.size, dtype=torch.uint8, device=gpu)
    itw.lib().itwSetStream(torch.cuda.current_stream().cuda_stream)
    surf = itw.RgbaSurface(d.data_ptr() + 63 * 96 * 4, 96, 64, -96 * 4)
    itw.lib().CompressBlocksBC3(C.byref(surf), d_out.data_ptr())
    torch.cuda.synchronize()
    assert first_mismatch(d_out.cpu().numpy(), want, 16) is None
    same = np.ascontiguousarray(np.broadcast_to(img[5:6], (8, 96, 4)))
    want = oracle.encode("bc1", same)
    out = np.zeros(want.size, dtype=np.uint8)
    surf = itw.RgbaSurface(img.ctypes.data + 5 * img.strides[0], 96, 8, 0)
    itw.lib().CompressBlocksBC1(C.byref(surf), out.ctypes.data)
    assert first_mismatch(out, want, 8) is None
