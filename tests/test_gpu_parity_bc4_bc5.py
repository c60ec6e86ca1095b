"""GPU parity, BC4_UNORM / BC5_UNORM: csrc/bc4_bc5.hip through the C ABI (include/itw_bc45.h) emits byte-identical
blocks to oracle/bc4_bc5.c (the restatement of the DirectXTex encoder the plugin calls for these formats,
IntelPlugin.cpp:271-273) and to the committed golden streams.  Bar: bit-exact -- the arithmetic is fp32 but the
output is integer codes and indices, and both sides evaluate every expression in the same order without contraction."""
import os

import numpy as np
import pytest

from conftest import first_mismatch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BPB = {"bc4": 8, "bc5": 16}


def gpu_encode(itw, gpu, fmt, img):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(img)).to(gpu)
    out = itw.compress(fmt, t)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.fixture(scope="module")
def golden45():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "golden_bc45.npz")))


@pytest.mark.parametrize("fmt", ["bc4", "bc5"])
@pytest.mark.parametrize("name", ["baboon", "edge_cases", "monkey_crop", "tiny"])
def test_golden_streams(itw, gpu, golden_inputs, golden45, fmt, name):
    from itw_amd import surfaces
    img = {"baboon": lambda: golden_inputs["baboon"], "edge_cases": surfaces.ldr_edge_cases,
           "monkey_crop": lambda: golden45["monkey_crop.input"], "tiny": lambda: golden45["tiny.input"]}[name]()
    got = gpu_encode(itw, gpu, fmt, img)
    want = golden45[f"{name}.{fmt}"]
    assert got.size == want.size
    assert first_mismatch(got, want, 8) is None, first_mismatch(got, want, 8)


@pytest.mark.parametrize("fmt", ["bc4", "bc5"])
@pytest.mark.parametrize("gen,h,w", [("ldr_smooth", 512, 512), ("ldr_uniform", 256, 512), ("ldr_smooth", 53, 101),
                                     ("ldr_uniform", 4, 4), ("ldr_uniform", 1, 1), ("ldr_uniform", 3, 2), ("ldr_uniform", 9, 263),
                                     ("ldr_smooth", 1023, 1030)])
def test_synthetic_vs_oracle(itw, gpu, oracle, fmt, gen, h, w):
    """Odd sizes included: DirectXTex keeps partial blocks (DirectXTexCompress.cpp:140-168)."""
    from itw_amd import surfaces
    H, W = (h + 3) // 4 * 4, (w + 3) // 4 * 4
    img = np.ascontiguousarray(getattr(surfaces, gen)(H, W)[:h, :w])
    got = gpu_encode(itw, gpu, fmt, img)
    want = oracle.encode_bc45(fmt, img)
    assert got.size == want.size == itw.block_count(fmt, w, h) * BPB[fmt]
    assert first_mismatch(got, want, 8) is None, first_mismatch(got, want, 8)


def _boundary_heavy(h, w, seed):
    """Content that drives the six-step codec and its corner cases: many exact 0 / 255 texels, flat blocks, blocks
    whose interior values collapse (fX == fY), two-level blocks."""
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    kind = rng.integers(0, 6, (h // 4, w // 4))
    k = np.repeat(np.repeat(kind, 4, axis=0), 4, axis=1)[..., None]
    base = np.repeat(np.repeat(rng.integers(0, 256, (h // 4, w // 4, 4), dtype=np.uint8), 4, axis=0), 4, axis=1)
    img = np.where(k == 1, base, img)                                                        # flat
    img = np.where(k == 2, np.where(img < 90, 0, np.where(img > 170, 255, base)), img)       # 0 / 255 / one interior value
    img = np.where(k == 3, np.where(img < 128, 0, 255), img)                                 # only boundary values
    img = np.where(k == 4, np.clip(base.astype(np.int32) + (img.astype(np.int32) % 7) - 3, 0, 255).astype(np.uint8), img)
    img = np.where(k == 5, np.where(img < 40, 0, img), img)                                  # noise with exact zeros
    return np.ascontiguousarray(img.astype(np.uint8))


@pytest.mark.parametrize("fmt", ["bc4", "bc5"])
def test_boundary_value_content(itw, gpu, oracle, fmt):
    img = _boundary_heavy(512, 512, 45)
    got = gpu_encode(itw, gpu, fmt, img)
    want = oracle.encode_bc45(fmt, img)
    assert first_mismatch(got, want, 8) is None, first_mismatch(got, want, 8)
    ends = want.reshape(-1, 8)[:, :2]
    assert (ends[:, 0] <= ends[:, 1]).mean() > 0.2 and (ends[:, 0] > ends[:, 1]).mean() > 0.2     # both codecs exercised


@pytest.mark.parametrize("fmt", ["bc4", "bc5"])
def test_host_pointers_and_strided_rows(itw, gpu, oracle, fmt):
    """The plugin passes host memory (IntelPlugin.cpp:271); rows may carry a pitch; the vector load path needs 16-byte
    alignment and must not be taken otherwise."""
    import torch
    from itw_amd import surfaces
    img = np.ascontiguousarray(surfaces.ldr_smooth(64, 72)[:61, :70])
    want = oracle.encode_bc45(fmt, img)
    assert first_mismatch(itw.compress_numpy(fmt, img), want, 8) is None
    wide = np.zeros((61, 75, 4), np.uint8)                        # pitch 300 B: rows not 16-byte aligned
    wide[:, 3:73] = img
    view = wide[:, 3:73]
    assert first_mismatch(itw.compress_numpy(fmt, view), want, 8) is None
    t = torch.from_numpy(wide).to(gpu)[:, 3:73]                   # device-resident, strided, base offset 12 B
    out = itw.compress(fmt, t)
    torch.cuda.synchronize()
    assert first_mismatch(out.cpu().numpy(), want, 8) is None


@pytest.mark.parametrize("fmt", ["bc4", "bc5"])
def test_full_size_4096(itw, gpu, oracle, fmt):
    """The bench surface (4096^2 synthetic RGBA8), whole surface compared."""
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(4096, 4096)
    got = gpu_encode(itw, gpu, fmt, img)
    want = oracle.encode_bc45(fmt, img)
    assert first_mismatch(got, want, 8) is None, first_mismatch(got, want, 8)


def test_uniform_noise_2048(itw, gpu, oracle):
    from itw_amd import surfaces
    img = surfaces.ldr_uniform(2048, 2048)
    got = gpu_encode(itw, gpu, "bc5", img)
    want = oracle.encode_bc45("bc5", img)
    assert first_mismatch(got, want, 8) is None, first_mismatch(got, want, 8)


# ---- round 4: FindClosestUNORM through the device-built run table -----------------------------------------------------------

def test_device_run_table_equals_the_table_of_the_search_as_written(itw, gpu, oracle):
    """The table the device builds once (bc45_build_index_table: the reference's search, as written, for all 65 536 endpoint
    pairs) is word for word the table derived on the CPU from the oracle's FindClosestUNORM over all 256^3 cases -- which
    tests/test_bc45_index_table.py pins to the reference's own function and proves lossless."""
    from _bc45_runs import runs_table
    want, max_runs = runs_table(oracle.bc4_find_closest_table())
    got = np.zeros((65536, 4), dtype=np.uint32)
    assert itw.test_lib().itwTestBc45IndexTable(got.ctypes.data) == 0     # hooks build (same sources, include/itw_test_hooks.h)
    assert max_runs == 8 and np.array_equal(got, want), np.flatnonzero((got != want).any(axis=1))[:8]


@pytest.mark.parametrize("fmt", ["bc4", "bc5"])
def test_every_texel_code_against_hand_picked_ramps(itw, gpu, oracle, fmt):
    """Blocks built so that the optimiser lands on many different endpoint pairs while the 16 texels sweep the code range:
    two texels pin the ramp's ends, the other fourteen walk through every code (incl. 0 and 255: the 6-step ramp)."""
    rng = np.random.default_rng(45)
    blocks = []
    for lo in list(range(0, 256, 7)) + [0, 1, 254, 255]:
        for hi in (lo, min(lo + 1, 255), min(lo + 3, 255), min(lo + 17, 255), min(lo + 90, 255), 255):
            t = rng.integers(min(lo, hi), max(lo, hi) + 1, size=16)
            t[rng.integers(0, 16)] = lo
            t[rng.integers(0, 16)] = hi
            blocks.append(t)
    blocks = np.array(blocks, dtype=np.uint8)                                    # (n, 16)
    n = blocks.shape[0]
    cols = 32
    rows = -(-n // cols)
    img = np.zeros((rows * 4, cols * 4, 4), dtype=np.uint8)
    for i, t in enumerate(blocks):
        by, bx = divmod(i, cols)
        img[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4, 0] = t.reshape(4, 4)
        img[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4, 1] = t[::-1].reshape(4, 4)
    got = gpu_encode(itw, gpu, fmt, img)
    want = oracle.encode(fmt, img).reshape(-1)
    assert first_mismatch(got, want, 8) is None, first_mismatch(got, want, 8)


@pytest.mark.parametrize("fmt", ["bc4", "bc5"])
def test_flat_and_two_level_blocks(itw, gpu, oracle, fmt):
    """red_0 == red_1 is where the interpolated levels sit an ulp off the endpoints and the search is not monotone (70 pairs):
    flat blocks of every code, and blocks of two adjacent codes."""
    img = np.zeros((8, 256 * 4, 4), dtype=np.uint8)
    for v in range(256):
        img[0:4, v * 4:v * 4 + 4, 0] = v
        img[0:4, v * 4:v * 4 + 4, 1] = 255 - v
        img[4:8, v * 4:v * 4 + 4, 0] = np.array([v, min(v + 1, 255)] * 8, dtype=np.uint8).reshape(4, 4)
        img[4:8, v * 4:v * 4 + 4, 1] = np.array([v, max(v - 1, 0)] * 8, dtype=np.uint8).reshape(4, 4)
    got = gpu_encode(itw, gpu, fmt, img)
    want = oracle.encode(fmt, img).reshape(-1)
    assert first_mismatch(got, want, 8) is None, first_mismatch(got, want, 8)


@pytest.mark.parametrize("fmt", ["bc4", "bc5"])
def test_whole_surface_walk_with_32_bit_offsets(itw, gpu, oracle, fmt):
    """Surfaces whose sides are multiples of 4 and that span < 2 GiB take the kernel's WHOLE walk: each lane keeps its block position and
    a 32-bit byte offset incrementally across the chunks its persistent workgroup visits.  Blocks are independent, so a 16384^2 surface
    tiled from a 4096^2 one (offsets up to 2^30) must encode to 16 copies of the small surface's stream; a strided view (row stride larger
    than the row, odd block counts) must equal its contiguous copy; and the small surface equals the oracle."""
    import torch
    from itw_amd import surfaces
    bpb = BPB[fmt]
    img = surfaces.ldr_smooth(1024, 1024)
    want = oracle.encode(fmt, img).reshape(-1)
    base = torch.from_numpy(img).to(gpu)
    small = itw.compress(fmt, base)
    torch.cuda.synchronize()
    assert first_mismatch(small.cpu().numpy(), want, 8) is None
    big = base.repeat(16, 16, 1).contiguous()                   # 16384 x 16384
    out = itw.compress(fmt, big).view(4096, 4096, bpb)
    torch.cuda.synchronize()
    s2 = small.view(256, 256, bpb)
    for i in (0, 3, 7, 15):
        for j in (0, 5, 15):
            assert torch.equal(out[256 * i:256 * (i + 1), 256 * j:256 * (j + 1)], s2), (i, j)
    sub = big[8:8 + 4 * 777, 16:16 + 4 * 333]
    assert torch.equal(itw.compress(fmt, sub), itw.compress(fmt, sub.contiguous()))
