"""GPU parity, BC6H: csrc/bc6h.hip (through the C ABI) vs the oracle (oracle/bc6h.c, restating
kernel.ispc:2039-3139) and the committed golden streams, for every quality profile.  Bar: bit-exact.
Includes adversarial input (random 16-bit patterns: negative halves, Inf, NaN, int-overflowing errors)."""
import numpy as np
import pytest

from conftest import first_mismatch

pytestmark = pytest.mark.gpu
ALL = ["veryfast", "fast", "basic", "slow", "veryslow"]


def gpu_encode(itw, gpu, img, prof):
    import torch
    t = torch.from_numpy(img.view(np.int16)).to(gpu)
    out = itw.compress("bc6h", t, prof)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("prof", ALL)
@pytest.mark.parametrize("name", ["monkey_hdr", "hdr_random_bits"])
def test_golden(itw, gpu, golden_inputs, golden_blocks, prof, name):
    got = gpu_encode(itw, gpu, golden_inputs[name], prof)
    want = golden_blocks[f"{name}.bc6h.{prof}"]
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


@pytest.mark.parametrize("prof", ["fast", "basic", "slow"])
@pytest.mark.parametrize("gen,h,w", [("hdr_smooth", 256, 256), ("hdr_random_bits", 128, 128), ("hdr_smooth", 36, 100)])
def test_synthetic_vs_oracle(itw, gpu, oracle, prof, gen, h, w):
    from itw_amd import surfaces
    img = getattr(surfaces, gen)(h, w)
    got = gpu_encode(itw, gpu, img, prof)
    want = oracle.encode_mt("bc6h", img, prof)
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


def test_special_values(itw, gpu, oracle):
    """Blocks of all-zero, all-max-finite (0x7BFF), Inf (0x7C00), NaN, sign bit set, and constant colour."""
    img = np.zeros((4, 4 * 8, 4), dtype=np.uint16)
    for i, v in enumerate([0x0000, 0x7BFF, 0x7C00, 0x7E00, 0x8000, 0xFFFF, 0x3C00, 0x0001]):
        img[:, 4 * i:4 * i + 4, :3] = v
    img[..., 3] = 0x3C00
    for prof in ALL:
        got = gpu_encode(itw, gpu, img, prof)
        want = oracle.encode("bc6h", img, prof)
        assert first_mismatch(got, want, 16) is None, (prof, first_mismatch(got, want, 16))


def test_host_pointer_and_strided(itw, gpu, oracle):
    import torch
    from itw_amd import surfaces
    img = surfaces.hdr_smooth(64, 64)
    want = oracle.encode("bc6h", img, "basic")
    assert (itw.compress_numpy("bc6h", img, "basic") == want).all()
    big = torch.zeros((64, 80, 4), dtype=torch.int16, device=gpu)
    big[:, 5:69] = torch.from_numpy(img.view(np.int16)).to(gpu)       # base offset 40 B, stride 640 B
    got = itw.compress("bc6h", big[:, 5:69], "basic")
    torch.cuda.synchronize()
    assert (got.cpu().numpy() == want).all()


def test_full_size_4096_properties(itw, gpu, oracle):
    """BASELINE configs[3] at full size: sampled bands bit-exact vs the oracle, periodicity of a tiled surface,
    and every block decodes (from-spec decoder) to within a few percent of its source."""
    from itw_amd import surfaces
    cell = surfaces.hdr_smooth(512, 512)
    img = surfaces.tile_to(cell, 4096, 4096)
    got = gpu_encode(itw, gpu, img, "slow").reshape(1024, 1024, 16)
    for y0 in (0, 2044, 4080):
        want = oracle.encode_mt("bc6h", img[y0:y0 + 16], "slow").reshape(4, 1024, 16)
        assert (got[y0 // 4:y0 // 4 + 4] == want).all(), y0
    tile = got[:128, :128]
    for ty in range(8):
        for tx in range(8):
            assert (got[ty * 128:(ty + 1) * 128, tx * 128:(tx + 1) * 128] == tile).all(), (ty, tx)
    dec, modes = oracle.decode("bc6h", np.ascontiguousarray(tile).reshape(-1), 512, 512)
    assert (modes >= 0).all()
    f = lambda a: a.astype(np.uint16).view(np.float16).astype(np.float64)
    rel = np.abs(f(dec) - f(cell[..., :3])) / np.maximum(f(cell[..., :3]), 1e-3)
    assert np.median(rel) < 0.10      # the 512-px cell of the synthetic field is busy: ~6 % median error at 8 bpp


@pytest.mark.parametrize("prof", ["fast", "slow"])
def test_full_size_i4_monkey_hdr_whole_surface(itw, gpu, oracle, golden_inputs, prof):
    """SURVEY 8(d) input I4 / BASELINE configs[3]: the reference's monkey-32bit.hdr (RGBE -> half, committed fixture) tiled
    to 4096 x 4096; all 1 048 576 blocks against the threaded oracle."""
    from itw_amd import surfaces
    img = surfaces.tile_to(golden_inputs["monkey_hdr"], 4096, 4096)
    got = gpu_encode(itw, gpu, img, prof)
    want = oracle.encode_mt("bc6h", img, prof).reshape(-1)
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


def test_random_settings_fuzz(itw, gpu, oracle):
    """bc6h_enc_settings is a caller-owned POD (ispc_texcomp.h:43-50): 96 random structs (mode gates, refine counts 0..4,
    thresholds 0..32) on smooth + adversarial content, bit-exact against the oracle."""
    from itw_amd import surfaces
    rng = np.random.default_rng(66)
    img = np.ascontiguousarray(np.concatenate([surfaces.hdr_smooth(32, 64), surfaces.hdr_random_bits(16, 64)], axis=0))
    for trial in range(96):
        s, so = itw.Bc6hSettings(), oracle.Bc6hSettings()
        vals = {"slow_mode": bool(rng.integers(0, 2)), "fast_mode": bool(rng.integers(0, 2)),
                "refineIterations_1p": int(rng.integers(0, 5)), "refineIterations_2p": int(rng.integers(0, 5)),
                "fastSkipTreshold": int(rng.choice([0, 1, 2, 4, 7, 16, 31, 32]))}
        for t in (s, so):
            for k, v in vals.items():
                setattr(t, k, v)
        got = gpu_encode(itw, gpu, img, s)
        want = oracle.encode("bc6h", img, so)
        assert first_mismatch(got, want, 16) is None, (trial, vals, first_mismatch(got, want, 16))
