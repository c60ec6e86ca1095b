"""GPU parity, BC7: csrc/bc7.hip (through the C ABI) vs the oracle (oracle/bc7.c, restating kernel.ispc:616-2037)
and the committed golden streams, for every quality profile.  Bar: bit-exact."""
import numpy as np
import pytest

from conftest import first_mismatch

pytestmark = pytest.mark.gpu

ALL = ["ultrafast", "veryfast", "fast", "basic", "slow",
       "alpha_ultrafast", "alpha_veryfast", "alpha_fast", "alpha_basic", "alpha_slow"]


def gpu_encode(itw, gpu, img, prof):
    import torch
    out = itw.compress("bc7", torch.from_numpy(img).to(gpu), prof)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("prof", ALL)
def test_golden_edge_cases(itw, gpu, golden_inputs, golden_blocks, prof):
    got = gpu_encode(itw, gpu, golden_inputs["edge_cases"], prof)
    want = golden_blocks[f"edge_cases.bc7.{prof}"]
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


@pytest.mark.parametrize("prof", ALL)
def test_golden_monkey(itw, gpu, golden_inputs, golden_blocks, prof):
    """220x220 photo with real alpha (reference sample image): every mode 0-7 occurs."""
    got = gpu_encode(itw, gpu, golden_inputs["monkey"], prof)
    want = golden_blocks[f"monkey.bc7.{prof}"]
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


@pytest.mark.parametrize("prof", ["veryfast", "basic", "slow", "alpha_basic"])
def test_golden_baboon(itw, gpu, golden_inputs, golden_blocks, prof):
    got = gpu_encode(itw, gpu, golden_inputs["baboon"], prof)
    want = golden_blocks[f"baboon.bc7.{prof}"]
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


@pytest.mark.parametrize("prof", ["basic", "slow", "alpha_slow"])
@pytest.mark.parametrize("gen,h,w", [("ldr_smooth", 256, 256), ("ldr_uniform", 128, 128), ("ldr_smooth", 36, 100)])
def test_synthetic_vs_oracle(itw, gpu, oracle, prof, gen, h, w):
    from itw_amd import surfaces
    img = getattr(surfaces, gen)(h, w)
    got = gpu_encode(itw, gpu, img, prof)
    want = oracle.encode_mt("bc7", img, prof)
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


def test_custom_settings_struct(itw, gpu, oracle):
    """Settings are a caller-owned POD (ispc_texcomp.h:27-41): odd but legal combinations must agree too --
    mode 7 enabled on an RGB profile, mode 2 only, unequal partition counts."""
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(64, 64)
    for tweak in ({"fastSkipTreshold_mode7": 5}, {"fastSkipTreshold_mode1": 0, "fastSkipTreshold_mode3": 7},
                  {"fastSkipTreshold_mode1": 20, "fastSkipTreshold_mode3": 3}, {"mode45_channel0": 2},
                  # 16 / 17: the boundary between the in-register top-16 ranking and the LDS-key ranking (bc7.hip RANKED 2 / 1)
                  {"fastSkipTreshold_mode1": 16, "fastSkipTreshold_mode3": 16, "fastSkipTreshold_mode7": 16},
                  {"fastSkipTreshold_mode1": 17, "fastSkipTreshold_mode3": 2, "fastSkipTreshold_mode7": 17},
                  {"fastSkipTreshold_mode1": 1, "fastSkipTreshold_mode3": 0, "fastSkipTreshold_mode7": 1}):
        s = itw.bc7_profile("basic")
        so = oracle.bc7_profile("basic")
        for k, v in tweak.items():
            setattr(s, k, v)
            setattr(so, k, v)
        s.refineIterations[7] = 1
        so.refineIterations[7] = 1
        got = gpu_encode(itw, gpu, img, s)
        want = oracle.encode("bc7", img, so)
        assert first_mismatch(got, want, 16) is None, (tweak, first_mismatch(got, want, 16))


def _posterised(h, w, levels, seed=5):
    """Few distinct colours per block: many shapes reach the same error, so the rank-key tie-break decides."""
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(h, w, seed=surfaces.SEED + seed)
    step = 256 // levels
    out = (img // step) * step + step // 2
    rng = np.random.default_rng(seed)
    flat = rng.random((h // 4, w // 4)) < 0.2                      # a fifth of the blocks completely flat
    blocks = out.reshape(h // 4, 4, w // 4, 4, 4).transpose(0, 2, 1, 3, 4).copy()      # [by][bx][y][x][c]
    blocks[flat] = blocks[flat][:, :1, :1, :]
    return np.ascontiguousarray(blocks.transpose(0, 2, 1, 3, 4).reshape(h, w, 4)).astype(np.uint8)


@pytest.mark.parametrize("prof", ["slow", "alpha_slow"])
@pytest.mark.parametrize("levels", [2, 4, 8])
def test_ties_between_shapes_follow_the_rank_order(itw, gpu, oracle, prof, levels):
    """Whole-table scans evaluate the PCA rank key only when two shapes tie (csrc/bc7.hip); the reference's ranked,
    strict-`<` scan must still be reproduced on content where ties are the rule."""
    img = _posterised(96, 128, levels)
    got = gpu_encode(itw, gpu, img, prof)
    want = oracle.encode_mt("bc7", img, prof)
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


def test_mixed_whole_table_and_ranked_lists(itw, gpu, oracle):
    """Thresholds on either side of 64 select different kernels (table-order scan vs ranked list): every mix agrees."""
    from itw_amd import surfaces
    img = np.concatenate([surfaces.ldr_smooth(32, 96), _posterised(32, 96, 4)], axis=0)
    for base, tweak in (("slow", {"fastSkipTreshold_mode3": 8}), ("slow", {"fastSkipTreshold_mode1": 0}),
                        ("slow", {"fastSkipTreshold_mode3": 0}), ("slow", {"fastSkipTreshold_mode1": 63, "fastSkipTreshold_mode3": 65}),
                        ("slow", {"fastSkipTreshold_mode7": 64}), ("alpha_slow", {"fastSkipTreshold_mode7": 1}),
                        ("alpha_slow", {"fastSkipTreshold_mode1": 1, "fastSkipTreshold_mode3": 64}), ("slow", {"skip_mode2": True}),
                        ("alpha_slow", {"channels": 3}), ("slow", {"channels": 4, "fastSkipTreshold_mode7": 64})):
        s = itw.bc7_profile(base)
        so = oracle.bc7_profile(base)
        for k, v in tweak.items():
            setattr(s, k, v)
            setattr(so, k, v)
        s.refineIterations[7] = 2
        so.refineIterations[7] = 2
        got = gpu_encode(itw, gpu, img, s)
        want = oracle.encode("bc7", img, so)
        assert first_mismatch(got, want, 16) is None, (base, tweak, first_mismatch(got, want, 16))


def test_every_subset_of_mode_families(itw, gpu, oracle):
    """mode_selection[4] switches the families {0,2} {1,3,7} {4,5} {6}; the families are separate kernels that hand the
    best error to each other, so every one of the 16 combinations must still equal the reference's single pass."""
    from itw_amd import surfaces
    img = np.concatenate([surfaces.ldr_smooth(16, 64), _posterised(16, 64, 4)], axis=0)
    for base in ("slow", "alpha_basic"):
        for bits in range(16):
            s = itw.bc7_profile(base)
            so = oracle.bc7_profile(base)
            for i in range(4):
                s.mode_selection[i] = bool(bits >> i & 1)
                so.mode_selection[i] = bool(bits >> i & 1)
            got = gpu_encode(itw, gpu, img, s)
            want = oracle.encode("bc7", img, so)
            assert first_mismatch(got, want, 16) is None, (base, bits, first_mismatch(got, want, 16))


def test_full_size_4096_slow_properties(itw, gpu, oracle):
    """BASELINE configs[2] at full size.  The scalar oracle needs minutes for 4096^2 'slow', so: (1) bands sampled
    across the surface are compared bit-exactly, (2) size-independent properties cover the rest -- blocks are
    independent, so a surface tiled from a 512x512 cell must produce the cell's block rows periodically, and every
    block must decode (from-spec decoder) close to its source."""
    import torch
    from itw_amd import surfaces
    cell = surfaces.ldr_smooth(512, 512)
    img = surfaces.tile_to(cell, 4096, 4096)
    got = gpu_encode(itw, gpu, img, "slow").reshape(1024, 1024, 16)
    # (1) oracle on sampled bands (4 block rows each, ~4k blocks per band)
    for y0 in (0, 1372, 4080):
        want = oracle.encode_mt("bc7", img[y0:y0 + 16], "slow").reshape(4, 1024, 16)
        assert (got[y0 // 4:y0 // 4 + 4] == want).all(), y0
    # (2a) periodicity: every 128x128-block tile equals the first
    tile = got[:128, :128]
    for ty in range(8):
        for tx in range(8):
            assert (got[ty * 128:(ty + 1) * 128, tx * 128:(tx + 1) * 128] == tile).all(), (ty, tx)
    # (2b) decode one tile with the from-spec decoder: valid modes, sane PSNR
    dec, modes = oracle.decode("bc7", np.ascontiguousarray(tile).reshape(-1), 512, 512)
    assert (modes >= 0).all()
    mse = np.mean((dec[..., :3].astype(np.float64) - cell[..., :3].astype(np.float64)) ** 2)
    assert 10 * np.log10(255 ** 2 / mse) > 30.0


def test_band_of_the_16k_configuration(itw, gpu):
    """BASELINE configs[4]: 16384^2 over 8 GPUs = a 16384 x 2048 band per GPU (band rule: 512 block rows each).  One such
    band on this GPU: tiled from a 512^2 cell it must reproduce the cell's block stream periodically (blocks are
    independent), which also exercises 2 M blocks per launch and the wide pitch."""
    import torch
    from itw_amd import surfaces
    y0, rows, off = itw.band_for_part(16384, 16384, "bc7", 3, 8)
    assert (y0, rows, off) == (3 * 2048, 2048, 3 * 512 * 4096 * 16)
    cell = surfaces.ldr_smooth(512, 512)
    band = surfaces.tile_to(cell, 2048, 16384)
    got = gpu_encode(itw, gpu, band, "slow").reshape(512, 4096, 16)
    tile = gpu_encode(itw, gpu, cell, "slow").reshape(128, 128, 16)
    for ty in range(4):
        for tx in range(32):
            assert (got[ty * 128:(ty + 1) * 128, tx * 128:(tx + 1) * 128] == tile).all(), (ty, tx)


@pytest.mark.parametrize("fmt,prof", [("bc7", "veryfast"), ("bc7", "alpha_basic"), ("bc6h", "fast")])
def test_host_pointers_chunked_overlap(itw, gpu, oracle, fmt, prof):
    """Large host-pointer calls of BC7 / BC6H are cut into runs of block rows whose uploads / downloads overlap the kernels
    (abi.hip compress(); 512 / 256 block rows per run by default, forced to four runs here with ITW_HOST_CHUNKS): unequal
    last run (130 and 131 block rows), pitched rows, a bottom-up (negative-stride) surface, and a device destination
    with a host source."""
    import ctypes as C
    import os
    import torch
    from itw_amd import surfaces
    os.environ["ITW_HOST_CHUNKS"] = "4"
    try:
        _chunked_cases(itw, gpu, oracle, fmt, prof, C, torch, surfaces)
    finally:
        del os.environ["ITW_HOST_CHUNKS"]


def _chunked_cases(itw, gpu, oracle, fmt, prof, C, torch, surfaces):
    for h, w in ((520, 64), (524, 36)):
        img = surfaces.hdr_smooth(h, w) if fmt == "bc6h" else surfaces.ldr_smooth(h, w)
        want = oracle.encode(fmt, img, prof)
        assert first_mismatch(itw.compress_numpy(fmt, img, prof), want, 16) is None, (h, w)
        wide = np.zeros((h, w + 5, 4), img.dtype)
        wide[:, 2:w + 2] = img
        assert first_mismatch(itw.compress_numpy(fmt, wide[:, 2:w + 2], prof), want, 16) is None, (h, w, "pitched")
    flipped = np.ascontiguousarray(img[::-1])
    want = oracle.encode(fmt, flipped, prof)
    out = np.zeros(want.size, dtype=np.uint8)
    surf = itw.RgbaSurface(img.ctypes.data + (h - 1) * img.strides[0], w, h, -img.strides[0])
    st = itw.bc6h_profile(prof) if fmt == "bc6h" else itw.bc7_profile(prof)
    fn = itw.lib().CompressBlocksBC6H if fmt == "bc6h" else itw.lib().CompressBlocksBC7
    fn(C.byref(surf), out.ctypes.data, C.byref(st))
    assert first_mismatch(out, want, 16) is None, "bottom-up"
    d_out = torch.zeros(want.size, dtype=torch.uint8, device=gpu)          # host source, device destination
    fn(C.byref(surf), d_out.data_ptr(), C.byref(st))
    torch.cuda.synchronize()
    assert first_mismatch(d_out.cpu().numpy(), want, 16) is None, "host -> device"


@pytest.mark.parametrize("prof", ["slow", "alpha_slow"])
def test_full_size_4096_bench_surface_whole(itw, gpu, oracle, prof):
    """The exact surface bench.py times (synthetic I3, 4096 x 4096, seed of rank 0): all 1 048 576 blocks against the
    threaded oracle (~15-20 s on the GPU box's 16 host cores)."""
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(4096, 4096)
    got = gpu_encode(itw, gpu, img, prof)
    want = oracle.encode_mt("bc7", img, prof).reshape(-1)
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


def test_whole_16384_surface_on_one_gpu(itw, gpu, oracle):
    """BASELINE configs[4]'s surface (16384 x 16384 = 16.8 M blocks, 1 GiB of texels) in ONE device-resident call: the
    288 GB of HBM make the single-GPU case legal, and it exercises every offset computation beyond 2^31 bits / 2^28
    bytes.  The surface is a 512 x 512 cell tiled on the device; blocks are independent, so every 128 x 128-block tile
    of the output must equal the oracle's stream of the cell."""
    import torch
    from itw_amd import surfaces
    cell = surfaces.ldr_smooth(512, 512)
    d_cell = torch.from_numpy(cell).to(gpu)
    img = d_cell.repeat(32, 32, 1)                                     # (16384, 16384, 4), 1 GiB
    assert img.shape == (16384, 16384, 4) and img.is_contiguous()
    for fmt, prof, bpb in (("bc7", "slow", 16), ("bc1", None, 8)):
        out = itw.compress(fmt, img, prof)
        torch.cuda.synchronize()
        want = torch.from_numpy(oracle.encode_mt(fmt, cell, prof).reshape(128, 128 * bpb)).to(gpu)
        got = out.view(32, 128, 32, 128 * bpb)                         # [tile_y][block_row][tile_x][bytes of 128 blocks]
        same = (got == want[None, :, None, :]).all(dim=3).all(dim=1)   # per tile
        assert bool(same.all()), (fmt, torch.nonzero(~same)[:4].tolist())
        del out, got


def test_whole_16384_surface_rgba_profile_block_list(itw, gpu, oracle):
    """The same 16.8 M-block surface under an RGBA profile with opaque and translucent blocks mixed per block: the alpha group's
    finish kernel compacts ~8 M block ids into the RGB list (atomic per wave, offsets beyond 2^23 entries) and the RGB scans walk
    it with gathered texel loads.  Tiled cell, so every tile of the output must equal the oracle's stream of the cell."""
    import torch
    from itw_amd import surfaces
    cell = surfaces.ldr_alpha_variant(surfaces.ldr_smooth(512, 512), "mixed")
    d_cell = torch.from_numpy(cell).to(gpu)
    img = d_cell.repeat(32, 32, 1)
    out = itw.compress("bc7", img, "alpha_basic")
    torch.cuda.synchronize()
    want = torch.from_numpy(oracle.encode_mt("bc7", cell, "alpha_basic").reshape(128, 128 * 16)).to(gpu)
    got = out.view(32, 128, 32, 128 * 16)
    same = (got == want[None, :, None, :]).all(dim=3).all(dim=1)
    assert bool(same.all()), torch.nonzero(~same)[:4].tolist()


def test_random_settings_fuzz(itw, gpu, oracle):
    """The settings struct is a caller-owned POD (ispc_texcomp.h:27-41) and any combination is legal input: 160 random
    structs -- mode families on/off, refine counts 0..5 per mode, thresholds 0 / at the 16|17 and 64 boundaries /
    above 64, every mode45_channel0, both channel counts -- on mixed content, bit-exact against the oracle."""
    from itw_amd import surfaces
    rng = np.random.default_rng(77)
    img = np.concatenate([surfaces.ldr_smooth(32, 64), surfaces.ldr_uniform(16, 64), _posterised(16, 64, 4)], axis=0)
    img = np.ascontiguousarray(img)
    thresholds = [0, 1, 2, 5, 12, 16, 17, 40, 63, 64, 70]      # negative counts index the tables out of bounds in the reference
    for trial in range(160):
        s, so = itw.Bc7Settings(), oracle.Bc7Settings()
        vals = {"skip_mode2": bool(rng.integers(0, 2)), "fastSkipTreshold_mode1": int(rng.choice(thresholds)),
                "fastSkipTreshold_mode3": int(rng.choice(thresholds)), "fastSkipTreshold_mode7": int(rng.choice(thresholds)),
                "mode45_channel0": int(rng.integers(0, 4)), "refineIterations_channel": int(rng.integers(0, 6)),
                "channels": int(rng.choice([3, 4]))}
        sel = [bool(rng.integers(0, 2)) for _ in range(4)]
        if not any(sel):
            sel[int(rng.integers(0, 4))] = True
        ref = [int(rng.integers(0, 6)) for _ in range(8)]
        for t in (s, so):
            for k, v in vals.items():
                setattr(t, k, v)
            for i in range(4):
                t.mode_selection[i] = sel[i]
            for i in range(8):
                t.refineIterations[i] = ref[i]
        got = gpu_encode(itw, gpu, img, s)
        want = oracle.encode("bc7", img, so)
        assert first_mismatch(got, want, 16) is None, (trial, vals, sel, ref, first_mismatch(got, want, 16))
