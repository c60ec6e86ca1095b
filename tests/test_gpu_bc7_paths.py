"""BC7 launch shapes (csrc/bc7.hip): DEEP (one lane per block, a launch pair per mode family) and WIDE (every family's scan
split over several waves, winners joined by an ordered argmin: lowest error, then the reference's strict-`<` list order,
kernel.ispc:1320, 1348, 1404-1409).  Both must emit the oracle's bytes on the same inputs -- the wide path is what the
plugin's 0x40000-pixel slices (IntelPlugin.cpp:851) and win32Threads' per-thread bands get by default."""
import numpy as np
import pytest

from conftest import first_mismatch

pytestmark = pytest.mark.gpu

ALL = ["ultrafast", "veryfast", "fast", "basic", "slow",
       "alpha_ultrafast", "alpha_veryfast", "alpha_fast", "alpha_basic", "alpha_slow"]


@pytest.fixture
def paths(itw):
    yield itw.set_bc7_path
    itw.set_bc7_path("auto")


def _encode(itw, gpu, img, prof):
    import torch
    out = itw.compress("bc7", torch.from_numpy(img).to(gpu), prof)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _posterised(h, w, levels, seed=5):
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(h, w, seed=surfaces.SEED + seed)
    step = 256 // levels
    return (img // step * step).astype(np.uint8)


@pytest.mark.parametrize("prof", ALL)
def test_both_paths_reproduce_the_golden_streams(itw, gpu, paths, golden_inputs, golden_blocks, prof):
    for name in ("monkey", "edge_cases"):
        want = golden_blocks[f"{name}.bc7.{prof}"]
        for path in ("wide", "deep"):
            paths(path)
            got = _encode(itw, gpu, golden_inputs[name], prof)
            assert first_mismatch(got, want, 16) is None, (name, path, first_mismatch(got, want, 16))


@pytest.mark.parametrize("prof", ["basic", "slow", "alpha_basic", "alpha_slow", "veryfast"])
@pytest.mark.parametrize("levels", [2, 4])
def test_tie_heavy_content_ordered_argmin(itw, gpu, paths, oracle, prof, levels):
    """Posterised blocks: many shapes reach the same error, so which part of a split scan holds the winner is decided by
    the tie rule (table index for modes 0/2, rank key for 1/3/7) -- evaluated at merge time in the wide path."""
    img = _posterised(128, 256, levels)
    img[..., 3] = _posterised(128, 256, levels, seed=9)[..., 0]
    want = oracle.encode_mt("bc7", img, prof)
    for path in ("wide", "deep"):
        paths(path)
        got = _encode(itw, gpu, img, prof)
        assert first_mismatch(got, want, 16) is None, (path, first_mismatch(got, want, 16))


@pytest.mark.parametrize("h,w", [(4, 4), (8, 36), (64, 64), (512, 512), (1024, 1024)])
def test_wide_path_at_every_split_width(itw, gpu, paths, oracle, h, w):
    """1 block .. 65 536 blocks: the number of parts per scan goes 16, 16, 16, 4, 1 (launch_bc7_wide); 512 x 512 is the
    plugin's slice."""
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(h, w, seed=surfaces.SEED + 31)
    for prof in ("slow", "alpha_slow", "basic"):
        if h * w > 512 * 512 and prof != "slow":
            continue
        want = oracle.encode_mt("bc7", img, prof)
        paths("wide")
        got = _encode(itw, gpu, img, prof)
        assert first_mismatch(got, want, 16) is None, (prof, first_mismatch(got, want, 16))


def test_custom_settings_on_both_paths(itw, gpu, paths, oracle):
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(64, 96, seed=surfaces.SEED + 77)
    img[..., 3] = surfaces.ldr_uniform(64, 96)[..., 0]
    for base, tweak in (("basic", {"fastSkipTreshold_mode7": 5}), ("basic", {"fastSkipTreshold_mode1": 0, "fastSkipTreshold_mode3": 7}),
                        ("alpha_basic", {"mode45_channel0": 2}), ("slow", {"skip_mode2": True}),
                        ("basic", {"fastSkipTreshold_mode1": 16, "fastSkipTreshold_mode3": 16, "fastSkipTreshold_mode7": 16}),
                        ("basic", {"fastSkipTreshold_mode1": 17, "fastSkipTreshold_mode3": 2, "fastSkipTreshold_mode7": 17}),   # deep only (LDS keys)
                        ("basic", {"fastSkipTreshold_mode1": 64, "fastSkipTreshold_mode3": 5}),                                # deep only
                        ("alpha_slow", {"fastSkipTreshold_mode7": 3}), ("alpha_basic", {"fastSkipTreshold_mode1": 1, "fastSkipTreshold_mode3": 1, "fastSkipTreshold_mode7": 1})):
        s, so = itw.bc7_profile(base), oracle.bc7_profile(base)
        for k, v in tweak.items():
            setattr(s, k, v)
            setattr(so, k, v)
        s.refineIterations[7] = so.refineIterations[7] = 2
        want = oracle.encode("bc7", img, so)
        for path in ("wide", "deep"):
            paths(path)
            got = _encode(itw, gpu, img, s)
            assert first_mismatch(got, want, 16) is None, (base, tweak, path, first_mismatch(got, want, 16))


# ---- BC6H slow profiles: one kernel vs split two-region scan (csrc/bc6h.hip, same switch) ------------------------------

def _encode6(itw, gpu, img, prof):
    import torch
    out = itw.compress("bc6h", torch.from_numpy(img).to(gpu), prof)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("prof", ["slow", "veryslow", "basic", "fast"])
def test_bc6h_both_shapes_reproduce_the_golden_streams(itw, gpu, paths, golden_inputs, golden_blocks, prof):
    for name in ("monkey_hdr", "hdr_random_bits"):
        want = golden_blocks[f"{name}.bc6h.{prof}"]
        for path in ("wide", "deep"):
            paths(path)
            got = _encode6(itw, gpu, golden_inputs[name], prof)
            assert first_mismatch(got, want, 16) is None, (name, path, first_mismatch(got, want, 16))


@pytest.mark.parametrize("h,w", [(4, 4), (8, 36), (64, 64), (256, 256), (512, 512)])
def test_bc6h_wide_at_every_split_width(itw, gpu, paths, oracle, h, w):
    from itw_amd import surfaces
    for gen in (surfaces.hdr_smooth, surfaces.hdr_random_bits):      # random bits: NaN / Inf halves and overflowing errors (S5)
        img = gen(h, w)
        want = oracle.encode_mt("bc6h", img, "slow")
        paths("wide")
        got = _encode6(itw, gpu, img, "slow")
        assert first_mismatch(got, want, 16) is None, (gen.__name__, first_mismatch(got, want, 16))


def test_bc6h_custom_list_lengths_on_both_shapes(itw, gpu, paths, oracle):
    from itw_amd import surfaces
    img = np.concatenate([surfaces.hdr_smooth(32, 64), surfaces.hdr_random_bits(32, 64)], axis=0)
    for n, r1, r2 in ((1, 0, 0), (2, 1, 1), (3, 2, 0), (7, 2, 2), (32, 1, 3), (40, 2, 2)):
        s, so = itw.bc6h_profile("slow"), oracle.bc6h_profile("slow")
        for o in (s, so):
            o.fastSkipTreshold, o.refineIterations_1p, o.refineIterations_2p = n, r1, r2
        want = oracle.encode("bc6h", img, so)
        for path in ("wide", "deep"):
            paths(path)
            got = _encode6(itw, gpu, img, s)
            assert first_mismatch(got, want, 16) is None, (n, r1, r2, path, first_mismatch(got, want, 16))


@pytest.mark.parametrize("path", ["deep", "wide"])
def test_random_settings_fuzz_on_each_shape(itw, gpu, paths, oracle, path):
    """The settings struct is a caller-owned POD: 96 random structs (mode families on/off, refine counts, thresholds at the
    16|17 and 64 boundaries, every mode45_channel0, both channel counts) forced through each launch shape -- "deep" is the
    fused two-launch path (or the per-family kernels when a ranked list is longer than 16 shapes), "wide" the split scans
    (falling back to deep where it does not support the settings)."""
    from itw_amd import surfaces
    rng = np.random.default_rng(2026 + (1 if path == "wide" else 0))
    img = np.ascontiguousarray(np.concatenate([surfaces.ldr_smooth(32, 64, seed=surfaces.SEED + 5), surfaces.ldr_uniform(16, 64),
                                               _posterised(16, 64, 4)], axis=0))
    img[..., 3] = surfaces.ldr_smooth(64, 64, seed=surfaces.SEED + 6)[..., 0]
    thresholds = [0, 1, 2, 5, 12, 16, 17, 40, 63, 64, 70]
    paths(path)
    for trial in range(96):
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
        got = _encode(itw, gpu, img, s)
        want = oracle.encode("bc7", img, so)
        assert first_mismatch(got, want, 16) is None, (path, trial, vals, sel, ref, first_mismatch(got, want, 16))


def test_bc6h_random_settings_on_each_shape(itw, gpu, paths, oracle):
    from itw_amd import surfaces
    rng = np.random.default_rng(606)
    img = np.ascontiguousarray(np.concatenate([surfaces.hdr_smooth(32, 64), surfaces.hdr_random_bits(32, 64)], axis=0))
    for trial in range(48):
        s, so = itw.Bc6hSettings(), oracle.Bc6hSettings()
        vals = {"slow_mode": bool(rng.integers(0, 2)), "fast_mode": bool(rng.integers(0, 2)), "refineIterations_1p": int(rng.integers(0, 4)),
                "refineIterations_2p": int(rng.integers(0, 4)), "fastSkipTreshold": int(rng.choice([0, 1, 2, 3, 8, 10, 31, 32, 33, 64]))}
        for t in (s, so):
            for k, v in vals.items():
                setattr(t, k, v)
        want = oracle.encode("bc6h", img, so)
        for path in ("wide", "deep"):
            paths(path)
            got = _encode6(itw, gpu, img, s)
            assert first_mismatch(got, want, 16) is None, (path, trial, vals, first_mismatch(got, want, 16))


@pytest.mark.parametrize("prof", ["alpha_fast", "alpha_basic", "alpha_slow"])
def test_alpha_first_order_and_skipped_rgb_modes(itw, gpu, paths, oracle, prof):
    """RGBA profiles in the fused shape run modes 7,4,5,6 first and skip modes 0-3 for waves (64 consecutive blocks) whose
    every block has sum (255 - alpha)^2 above the alpha modes' error (bc7_finish_all, kernel.ispc:1267-1277, 1356).  Rows of
    blocks here are built per wave: opaque (nothing skipped, RGB modes win or tie), translucent (whole waves skipped),
    alpha 254 speckles (opaque-error of a few units against alpha-mode errors of the same size: ties and near ties), and
    waves mixing all of them block by block."""
    from itw_amd import surfaces
    h, w = 64, 1024                                           # 256 blocks per block row = 4 waves; 16 block rows
    rng = np.random.default_rng(77)
    img = surfaces.ldr_smooth(h, w, seed=surfaces.SEED + 21).copy()
    img[16:32] = _posterised(16, w, 4, seed=3)                # flat colours: zero-error fits, exact ties between the groups
    a = np.full((h, w), 255, np.uint8)
    a[:, 256:512] = rng.integers(90, 170, (h, 256))           # translucent waves
    speck = rng.random((h, 256)) < 0.08
    a[:, 512:768] = np.where(speck, 254, 255)                 # nearly opaque
    blk = rng.integers(0, 4, (h // 4, 64))                    # mixed wave: per block opaque / 254 / smooth ramp / noise
    kinds = np.repeat(np.repeat(blk, 4, 0), 4, 1)
    ramp = np.broadcast_to(np.linspace(0, 255, 256).astype(np.uint8), (h, 256))
    a[:, 768:] = np.select([kinds == 0, kinds == 1, kinds == 2], [255, 254, ramp], rng.integers(0, 256, (h, 256)))
    img[..., 3] = a
    want = oracle.encode_mt("bc7", img, prof)
    for path in ("deep", "wide"):
        paths(path)
        got = _encode(itw, gpu, img, prof)
        assert first_mismatch(got, want, 16) is None, (path, first_mismatch(got, want, 16))
    modes = np.array([int(b[0]).bit_length() and (int(b[0]) & -int(b[0])).bit_length() - 1 for b in want.reshape(-1, 16)])
    assert (modes <= 3).any() and (modes >= 4).any()          # both groups of modes win somewhere


@pytest.mark.parametrize("opaque_blocks", [0, 1, 7, 255, 256, 257, 1000, 4096])
def test_rgb_block_list_of_every_length(itw, gpu, paths, oracle, opaque_blocks):
    """Round 3: under an RGBA profile bc7_finish_all<1> compacts the blocks where an RGB mode can still win into a list
    (atomic per wave, any order) and the RGB scans / finish<2> walk the list.  A 256 x 256 surface (4 096 blocks) whose alpha is
    far from 255 except in `opaque_blocks` blocks scattered at random: list lengths 0 (every RGB workgroup leaves at once), 1, less
    than a wave, one workgroup exactly, one over, many, all -- same bytes as the oracle each time, and twice in a row (the counter
    is reset per call)."""
    from itw_amd import surfaces
    rng = np.random.default_rng(1000 + opaque_blocks)
    img = surfaces.ldr_smooth(256, 256, seed=surfaces.SEED + 41).copy()
    img[..., 3] = rng.integers(40, 120, (256, 256))
    chosen = rng.permutation(4096)[:opaque_blocks]
    for b in chosen:
        y, x = divmod(int(b), 64)
        img[4 * y:4 * y + 4, 4 * x:4 * x + 4, 3] = 255
    want = oracle.encode_mt("bc7", img, "alpha_basic")
    paths("deep")                                              # the fused launch shape at this size
    for _ in range(2):
        got = _encode(itw, gpu, img, "alpha_basic")
        assert first_mismatch(got, want, 16) is None, (opaque_blocks, first_mismatch(got, want, 16))
    if opaque_blocks == 4096:
        modes = np.array([(int(b[0]) & -int(b[0])).bit_length() - 1 for b in want.reshape(-1, 16)])
        assert (modes <= 3).any()                              # the RGB group does win on opaque blocks


@pytest.mark.parametrize("prof", ["slow", "alpha_basic", "alpha_slow"])
def test_unaligned_device_surfaces_take_the_dword_kernels(itw, gpu, paths, oracle, prof):
    """Base pointer 12 bytes past a 16-byte boundary, rows 1 168 bytes apart (rgba_surface.stride is free): the VEC16 = false
    instantiations of every scan / finish kernel, both launch shapes, incl. the two-phase RGBA order."""
    import torch
    from itw_amd import surfaces
    img = surfaces.ldr_smooth(64, 256, seed=surfaces.SEED + 33).copy()
    img[:, 128:, 3] = 255                                      # a translucent and an opaque half: skipped and unskipped waves
    big = torch.zeros((64, 292, 4), dtype=torch.uint8, device=gpu)
    big[:, 3:259] = torch.from_numpy(img).to(gpu)
    view = big[:, 3:259]
    assert view.data_ptr() % 16 == 12 and view.stride(0) == 1168
    want = oracle.encode_mt("bc7", img, prof)
    for path in ("deep", "wide"):
        paths(path)
        got = itw.compress("bc7", view, prof)
        torch.cuda.synchronize()
        assert first_mismatch(got.cpu().numpy(), want, 16) is None, (path, first_mismatch(got.cpu().numpy(), want, 16))
