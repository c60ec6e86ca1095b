"""The device side of the bounded BC7 mode order (csrc/bc7.hip bc7_finish_all phases 3-5, csrc/bc7_exact.hpp two_subset_bound):
the bound as the GPU computes it never exceeds the oracle's error of any shape (kernel.ispc:1279-1297) and agrees with its CPU
restatement (oracle/bc7_bound.c); content built to hit the order's tie rules encodes to the oracle's bytes."""
import ctypes as C

import numpy as np
import pytest

from conftest import first_mismatch
from test_bc7_bound import sample_images, planar_blocks, lib as oracle_lib

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,img", list(sample_images()), ids=[n for n, _ in sample_images()])
def test_device_bound_is_below_every_oracle_error_and_matches_its_restatement(itw, gpu, name, img):
    import torch
    h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
    img = np.ascontiguousarray(img[:h, :w])
    got = itw.bc7_two_subset_bounds(torch.from_numpy(img).to(gpu)).cpu().numpy().astype(np.float64)
    L = oracle_lib()
    blocks = planar_blocks(img)
    assert got.shape == (blocks.shape[0], 64)
    err = np.zeros(64, np.float32)
    key = np.zeros(64, np.int32)
    for b in range(blocks.shape[0]):
        cpu = np.array([L.oracle_bc7_two_subset_bound(blocks[b].ctypes.data, p) for p in range(64)], dtype=np.float64)
        # v_rcp_f32 / v_sqrt_f32 against 1/x / sqrtf, and rcp-of-count constants against 1/n: a few ulp, far inside the bound's margins
        assert np.allclose(got[b], cpu, rtol=2e-4, atol=0.02), (name, b, int(np.argmax(np.abs(got[b] - cpu))), got[b].max(), cpu.max())
        for mode in (1, 3):
            L.oracle_bc7_part_fast_errors(blocks[b].ctypes.data, mode, err.ctypes.data, key.ctypes.data)
            assert (got[b] <= err.astype(np.float64)).all(), (name, b, mode, float((got[b] - err).max()))


def _encode(itw, gpu, img, prof):
    import torch
    out = itw.compress("bc7", torch.from_numpy(img).to(gpu), prof)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.fixture
def deep(itw):
    itw.set_bc7_path("deep")            # the fused shape (whole surfaces): where the bounded order lives
    yield
    itw.set_bc7_path("auto")


@pytest.mark.parametrize("prof", ["slow", "alpha_slow"])
def test_bounded_order_tie_rules(itw, gpu, oracle, deep, prof):
    """Few-level content: many blocks reach the same error (often 0) in several modes, so who keeps the block is decided by the
    reference's order 0,2,1,3,7,4,5,6 with strict `<` -- modes 1/3 must lose ties against 0/2 and win them against 4/5/6/7."""
    from itw_amd import surfaces
    rng = np.random.default_rng(11)
    tiles = []
    for levels in (2, 3, 4, 8):
        base = surfaces.ldr_smooth(64, 256, seed=surfaces.SEED + levels)
        step = 256 // levels
        tiles.append((base // step * step).astype(np.uint8))
    two = np.zeros((64, 256, 4), np.uint8)                # two flat colours per block along every two-subset shape boundary-ish pattern
    for by in range(16):
        for bx in range(64):
            a, b = rng.integers(0, 256, 4), rng.integers(0, 256, 4)
            m = rng.integers(0, 2, (4, 4)).astype(bool)
            blk = np.where(m[..., None], a[None, None, :], b[None, None, :])
            two[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4] = blk
    tiles.append(two)
    img = np.concatenate(tiles, axis=0)
    if prof == "slow":
        img[..., 3] = 255
    else:
        img[::8, :, 3] = 255                              # opaque and translucent blocks mixed
    want = oracle.encode_mt("bc7", img, prof)
    got = _encode(itw, gpu, img, prof)
    assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


@pytest.mark.parametrize("prof", ["slow", "alpha_slow"])
def test_bounded_order_on_noisy_and_natural_content(itw, gpu, oracle, deep, golden_inputs, prof):
    """noisy content: most blocks never reach modes 1/3 (the list is short); natural content: nearly all do"""
    from itw_amd import surfaces
    noisy = surfaces.ldr_smooth(256, 256, seed=surfaces.SEED + 41)
    nat = np.ascontiguousarray(golden_inputs["baboon"])
    for img in (noisy, nat):
        img = img.copy()
        if prof == "alpha_slow":
            img[64:192, 64:192, 3] = 255
        want = oracle.encode_mt("bc7", img, prof)
        got = _encode(itw, gpu, img, prof)
        assert first_mismatch(got, want, 16) is None, first_mismatch(got, want, 16)


# ---- round 5: the pilot that picks the mode order per call, the compact texel buffer of the list scans, mode 6 skipped by its bound -------

def _mixed_content(golden_inputs, h, w):
    """noise over smooth fields (few blocks need modes 1/3), a photograph (nearly all do), few-level content (ties between the modes) and
    flat / two-colour blocks, side by side in bands so that the pilot's sample (one 256-block chunk in 16) sees all of them"""
    from itw_amd import surfaces
    rng = np.random.default_rng(5)
    bab = golden_inputs["baboon"]
    img = surfaces.ldr_smooth(h, w, seed=surfaces.SEED + 77).copy()
    q = h // 4
    nat = np.tile(bab, (-(-q // bab.shape[0]), -(-w // bab.shape[1]), 1))[:q, :w]
    img[q:2 * q] = nat
    img[2 * q:3 * q] = (img[2 * q:3 * q] // 64 * 64)
    for y in range(3 * q, h - 3, 4):
        for x in range(0, w - 3, 4):
            a, b = rng.integers(0, 256, 4), rng.integers(0, 256, 4)
            m = rng.integers(0, 2, (4, 4)).astype(bool) if (x // 4) % 3 else np.zeros((4, 4), bool)
            img[y:y + 4, x:x + 4] = np.where(m[..., None], a[None, None, :], b[None, None, :])
    img[..., 3] = 255
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("pilot", [-1, 0, 100, None])
def test_pilot_verdicts_and_compact_lists_emit_the_oracles_bytes(itw, gpu, oracle, deep, golden_inputs, pilot):
    """`slow` through the fused shape with the pilot forced either way (0: the rest of the surface takes the reference's order, 100: the
    bounded order), off (-1: round 4's whole-call bounded order) and at its default threshold: the same bytes as the oracle.  The surface
    has a chunk count that is not a multiple of the pilot's period and a partial last chunk (604 x 516: 151 x 129 blocks = 76.1 chunks)."""
    img = _mixed_content(golden_inputs, 516, 604)
    want = oracle.encode_mt("bc7", img, "slow")
    itw.set_bc7_pilot(pilot)
    try:
        got = _encode(itw, gpu, img, "slow")
    finally:
        itw.set_bc7_pilot(None)
    assert first_mismatch(got, want, 16) is None, (pilot, first_mismatch(got, want, 16))


def test_pilot_on_a_strided_surface_and_repeated_calls(itw, gpu, oracle, deep, golden_inputs):
    """rows of the device surface further apart than their texels, both verdicts back to back on one stream (the workspace, the lists'
    counters and the pilot's word are reused from call to call)"""
    import torch
    img = _mixed_content(golden_inputs, 260, 1028)
    want = oracle.encode_mt("bc7", img, "slow")
    wide = torch.zeros((260, 1100, 4), dtype=torch.uint8, device=gpu)
    wide[:, :1028] = torch.from_numpy(img).to(gpu)
    view = wide[:, :1028]
    try:
        for pilot in (0, 100, 0, None, -1):
            itw.set_bc7_pilot(pilot)
            out = itw.compress("bc7", view, "slow")
            torch.cuda.synchronize()
            assert first_mismatch(out.cpu().numpy(), want, 16) is None, pilot
    finally:
        itw.set_bc7_pilot(None)


def test_mode_6_skipped_by_its_bound_changes_nothing(itw, gpu, oracle, deep):
    """content where mode 6 wins most blocks (smooth one-line gradients), content where its bound rules it out everywhere (noise) and flat
    blocks (error 0 everywhere): `slow` and a custom struct with modes 0/2 + 6 only"""
    from itw_amd import surfaces
    rng = np.random.default_rng(9)
    g = np.linspace(0, 255, 256)
    grad = np.stack([np.tile(g, (64, 1)), np.tile(g[::-1], (64, 1)), np.tile(g * 0.5 + 20, (64, 1)), np.full((64, 256), 255.0)], axis=2)
    noise = rng.integers(0, 256, (64, 256, 4))
    flat = np.tile(rng.integers(0, 256, (1, 64, 1, 4)).repeat(4, axis=1).reshape(1, 256, 4), (64, 1, 1))
    img = np.concatenate([grad, noise, flat, surfaces.ldr_smooth(64, 256)], axis=0).astype(np.uint8)
    img[..., 3] = 255
    img = np.ascontiguousarray(img)
    for prof in ("slow", None):
        if prof is None:
            s = itw.bc7_profile("slow")
            s.mode_selection[1] = 0; s.mode_selection[2] = 0
            o = oracle.bc7_profile("slow")
            o.mode_selection[1] = 0; o.mode_selection[2] = 0
        else:
            s, o = prof, prof
        want = oracle.encode("bc7", img, o)
        got = _encode(itw, gpu, img, s)
        assert first_mismatch(got, want, 16) is None, (prof, first_mismatch(got, want, 16))


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_pilot_and_bands_under_random_settings_in_the_bounded_territory(itw, gpu, oracle, deep, golden_inputs, seed):
    """random settings structs that keep the bounded order's conditions (both multi-subset groups on, every two-subset shape scanned, no mode 7
    or -- RGBA -- mode 7 scanned in full) but vary everything else (refinement counts incl. 0, mode 2 on / off, modes 4/5 and 6 on / off, the first
    mode-4/5 rotation, RGB / RGBA), on a surface large enough for two bands and the pilot, with the pilot's verdict forced either way by turns"""
    rng = np.random.default_rng(seed)
    img = _mixed_content(golden_inputs, 516, 604)
    rgba = bool(seed % 3 == 0)
    if rgba:
        img = img.copy()
        img[::3, :, 3] = rng.integers(0, 256, img[::3, :, 3].shape)
    s, o = itw.bc7_profile("alpha_slow" if rgba else "slow"), oracle.bc7_profile("alpha_slow" if rgba else "slow")
    for t in (s, o):
        t.skip_mode2 = int(seed % 2)
        t.mode_selection[2] = int(seed % 4 != 1)
        t.mode_selection[3] = int(seed % 4 != 2)
        t.mode45_channel0 = seed % 3 if not rgba else seed % 4
    its = rng.integers(0, 5, 8)
    ch = int(rng.integers(0, 5))
    for t in (s, o):
        for i in range(8):
            t.refineIterations[i] = int(its[i])
        t.refineIterations_channel = ch
    want = oracle.encode_mt("bc7", img, o)
    for pilot in ((0, 100) if seed % 2 else (100, 0)):
        itw.set_bc7_pilot(pilot)
        try:
            got = _encode(itw, gpu, img, s)
        finally:
            itw.set_bc7_pilot(None)
        assert first_mismatch(got, want, 16) is None, (seed, pilot, first_mismatch(got, want, 16))
