"""The bounded BC7 mode order rests on ONE property: for every block and every two-subset shape, two_subset_bound(shape) never exceeds the
error of any mode 1 / 3 encoding of that shape.  CPU side: the bound's restatement (oracle/bc7_bound.c, the same float operations as
csrc/bc7_exact.hpp) against the oracle's own errors -- bc7_enc_mode01237_part_fast for every shape (kernel.ispc:1279-1297), the refined
result of a modes-1/3-only encode (:1329-1362), and brute-force palettes from arbitrary endpoints.  The device function is compared with
the restatement in tests/test_gpu_bc7_bound.py."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
from oracle import pyoracle  # noqa: E402
from itw_amd import surfaces  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def planar_blocks(img):
    h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
    t = img[:h, :w].astype(np.float32).reshape(h // 4, 4, w // 4, 4, 4).transpose(0, 2, 4, 1, 3).reshape(-1, 64)   # [block][16 c + k]
    return np.ascontiguousarray(t)


def adversarial_blocks(rng):
    """flat, two-level, checkerboards, extreme contrast, single outliers, noise of several amplitudes, ramps"""
    out = []
    for v in (0, 1, 127, 254, 255):
        out.append(np.full((4, 4, 4), v, np.uint8))
    for a, b in ((0, 255), (0, 1), (100, 101), (17, 240)):
        c = np.zeros((4, 4, 4), np.uint8)
        c[...] = a
        c[::2, ::2] = b
        c[1::2, 1::2] = b
        out.append(c)
        c = np.full((4, 4, 4), a, np.uint8)
        c[2, 3] = b
        out.append(c)
        c = np.full((4, 4, 4), a, np.uint8)
        c[:, 2:] = b
        out.append(c)
        c = np.full((4, 4, 4), a, np.uint8)
        c[:2, :, 0] = b
        c[:, :2, 1] = b
        c[::2, :, 2] = b
        out.append(c)
    for amp in (1, 2, 4, 16, 64, 255):
        for _ in range(40):
            base = rng.integers(0, 256, 3)
            c = np.clip(base[None, None, :] + rng.integers(-amp, amp + 1, (4, 4, 3)), 0, 255).astype(np.uint8)
            out.append(np.concatenate([c, np.full((4, 4, 1), 255, np.uint8)], axis=2))
    for _ in range(40):
        g = np.linspace(0, 1, 4)
        d0, d1 = rng.integers(-80, 81, 3), rng.integers(-80, 81, 3)
        c = rng.integers(60, 196, 3)[None, None, :] + g[:, None, None] * d0[None, None, :] + g[None, :, None] * d1[None, None, :] + rng.normal(0, 3, (4, 4, 3))
        out.append(np.concatenate([np.clip(c, 0, 255).astype(np.uint8), np.full((4, 4, 1), 255, np.uint8)], axis=2))
    for step, d in ((17, (1, 1, 1)), (17, (1, 0, 0)), (8, (1, 2, 0)), (5, (3, 1, 2)), (1, (1, 1, 0)), (17, (1, -1, 0))):
        # exactly collinear texels (residual 0, large trace): the bound has to come out as 0, not as rounding noise
        k = np.arange(16).reshape(4, 4, 1) * step
        c = np.clip(np.where(np.array(d) < 0, 255, 0)[None, None, :] + k * np.array(d)[None, None, :], 0, 255)
        for perm in range(3):
            cc = rng.permuted(c.reshape(16, 3), axis=0).reshape(4, 4, 3) if perm else c
            out.append(np.concatenate([cc.astype(np.uint8), np.full((4, 4, 1), 255, np.uint8)], axis=2))
    img = np.concatenate(out, axis=1)
    return np.ascontiguousarray(img)


def sample_images():
    rng = np.random.default_rng(20260927)
    z = np.load(os.path.join(GOLDEN, "inputs.npz"))
    z2 = np.load(os.path.join(GOLDEN, "samples2.npz"))
    yield "adversarial", adversarial_blocks(rng)
    yield "ldr_smooth", surfaces.ldr_smooth(128, 128)
    yield "baboon", np.ascontiguousarray(z["baboon"][64:160, 64:160])
    yield "monkey", np.ascontiguousarray(z["monkey"][40:104, 40:104])
    yield "edge_cases", z["edge_cases"]
    yield "colors260k", np.ascontiguousarray(z2["colors260k"][:64, :64])
    yield "normals", np.ascontiguousarray(z2["normals"][96:160, 96:160])
    yield "landscape", np.ascontiguousarray(z2["landscape_detail"][:96, :96])


def lib():
    L = pyoracle.lib()
    L.oracle_bc7_two_subset_bound.argtypes = [C.c_void_p, C.c_int]
    L.oracle_bc7_two_subset_bound.restype = C.c_float
    L.oracle_bc7_part_fast_errors.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.oracle_bc7_part_fast_errors.restype = None
    return L


def bounds_of(L, block):
    return np.array([L.oracle_bc7_two_subset_bound(block.ctypes.data, p) for p in range(64)], dtype=np.float64)


@pytest.mark.parametrize("name,img", list(sample_images()), ids=[n for n, _ in sample_images()])
def test_bound_never_exceeds_the_oracles_error_of_any_shape(name, img):
    L = lib()
    blocks = planar_blocks(img)
    err = np.zeros(64, np.float32)
    key = np.zeros(64, np.int32)
    only13 = pyoracle.bc7_profile("slow")
    only13.mode_selection[0] = 0; only13.mode_selection[2] = 0; only13.mode_selection[3] = 0
    data = (C.c_uint32 * 4)()
    e = C.c_float()
    L.oracle_bc7_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_bc7_block.restype = None
    tight = 0
    for b in range(blocks.shape[0]):
        lb = bounds_of(L, blocks[b])
        assert (lb >= 0).all() and np.isfinite(lb).all()
        for mode in (1, 3):
            L.oracle_bc7_part_fast_errors(blocks[b].ctypes.data, mode, err.ctypes.data, key.ctypes.data)
            assert (lb <= err.astype(np.float64)).all(), (name, b, mode, int(np.argmax(lb - err)), float((lb - err).max()))
        # the refined winner of a modes-1/3-only encode is an encoding of SOME shape: it cannot get below the smallest bound
        L.oracle_bc7_block(blocks[b].ctypes.data, C.byref(only13), data, C.byref(e))
        assert lb.min() <= e.value, (name, b, float(lb.min()), e.value)
        tight += lb.min() > 0.5 * e.value
    assert blocks.shape[0] > 0
    if name == "ldr_smooth":
        assert tight > 0.5 * blocks.shape[0]          # on noisy content the bound is within a factor of two of what the modes achieve


def test_bound_never_exceeds_brute_force_palettes_from_arbitrary_endpoints():
    """any integer endpoints, both index widths, best level per texel: still above the bound (the property needs no optimality)"""
    rng = np.random.default_rng(7)
    L = lib()
    weights = {2: np.array([0, 21, 43, 64]), 3: np.array([0, 9, 18, 27, 37, 46, 55, 64])}
    import re
    t = open(os.path.join(ROOT, "oracle", "bc7_tables.h")).read()
    m = re.search(r"BCN_SUBSET_MASKS\[128\]\s*=\s*\{([^}]*)\}", t)
    masks = [int(x, 16) & 0xffff for x in re.findall(r"0x([0-9a-fA-F]+)u", m.group(1))][:64]
    img = np.concatenate([adversarial_blocks(rng)[:, :400], surfaces.ldr_smooth(4, 400)], axis=1)
    blocks = planar_blocks(img)
    worst = np.inf
    for b in range(blocks.shape[0]):
        lb = bounds_of(L, blocks[b])
        tex = blocks[b].reshape(4, 16)[:3].T.astype(np.int64)              # [texel][rgb]
        for shape in rng.choice(64, 6, replace=False):
            in0 = np.array([(masks[shape] >> k) & 1 for k in range(16)], bool)
            for bits in (2, 3):
                for trial in range(6):
                    total = 0
                    for sel in (in0, ~in0):
                        pts = tex[sel]
                        if trial < 3:                                      # endpoints near the subset's own extremes: the competitive ones
                            a = pts[rng.integers(len(pts))] + rng.integers(-3, 4, 3)
                            c = pts[rng.integers(len(pts))] + rng.integers(-3, 4, 3)
                        else:
                            a, c = rng.integers(0, 256, 3), rng.integers(0, 256, 3)
                        a, c = np.clip(a, 0, 255), np.clip(c, 0, 255)
                        w = weights[bits][:, None]
                        pal = ((64 - w) * a[None, :] + w * c[None, :] + 32) >> 6
                        d = ((pts[:, None, :] - pal[None, :, :]) ** 2).sum(axis=2).min(axis=1)
                        total += int(d.sum())
                    assert lb[shape] <= total, (b, shape, bits, trial, lb[shape], total)
                    worst = min(worst, total - lb[shape])
    assert worst >= 0


def test_list_scan_share_mapping_covers_every_chunk_and_share_exactly_once():
    """The bounded order's list scans (csrc/bc7.hip: list_scan_parts and the split branch of bc7_scan_all) cut each listed block's 64
    shapes into `parts` strided shares and map workgroup w to (chunk, share) so that a chunk's shares sit on one XCD (w % 8).  Restated
    here: for every list length the workgroups of the launch (grid >= ceil(nblocks / 256) rounded up to 8) cover every (chunk, share)
    exactly once, every share's winners fit the mode's winner row (parts x listed <= nblocks), and the shares partition the 64 shapes."""
    TPB, MAXP = 256, 8

    def parts_of(count, nblocks):
        chunks = ((count + TPB - 1) // TPB + 7) & ~7
        if chunks <= 0:
            return 1
        return max(1, min(MAXP, nblocks // (chunks * TPB)))

    for nblocks in (1, 9, 255, 256, 257, 4096, 65536, 262145, 1 << 20):
        nchunks = (nblocks + TPB - 1) // TPB
        grid = (nchunks + 7) // 8 * 8
        counts = sorted({0, 1, 2, 255, 256, 257, nblocks // 7, nblocks // 3, nblocks // 2, nblocks - 1, nblocks} & set(range(nblocks + 1)))
        for count in counts:
            parts = parts_of(count, nblocks)
            assert parts * count <= nblocks or parts == 1
            seen = {}
            for w in range(grid):
                chunk, part = (w // (8 * parts)) * 8 + (w & 7), (w >> 3) % parts
                if chunk * TPB >= count:
                    continue
                assert (chunk, part) not in seen
                seen[(chunk, part)] = w
            need = {(c, p) for c in range((count + TPB - 1) // TPB) for p in range(parts)}
            assert set(seen) == need, (nblocks, count, parts)
            for (c, p), w in seen.items():
                assert w % 8 == seen[(c, 0)] % 8                     # a chunk's shares: the same XCD
            shapes = sorted(s for p in range(parts) for s in range(p, 64, parts))
            assert shapes == list(range(64))


def test_bound_on_twenty_thousand_random_blocks_of_five_kinds():
    """seeded random blocks (uniform bytes, two colours + noise, planar gradients + noise, low-amplitude noise around a colour, one-channel
    noise): bound <= part_fast error of every shape for modes 1 and 3"""
    rng = np.random.default_rng(4242)
    n = 4000
    kinds = []
    kinds.append(rng.integers(0, 256, (n, 16, 3)))
    a, b = rng.integers(0, 256, (n, 1, 3)), rng.integers(0, 256, (n, 1, 3))
    pick = rng.integers(0, 2, (n, 16, 1))
    kinds.append(np.where(pick == 1, a, b) + rng.integers(-3, 4, (n, 16, 3)))
    gx, gy = np.meshgrid(np.arange(4), np.arange(4))
    g = (gx.reshape(1, 16, 1) * rng.integers(-40, 41, (n, 1, 3)) + gy.reshape(1, 16, 1) * rng.integers(-40, 41, (n, 1, 3)))
    kinds.append(rng.integers(40, 216, (n, 1, 3)) + g + rng.integers(-2, 3, (n, 16, 3)))
    kinds.append(rng.integers(0, 256, (n, 1, 3)) + rng.integers(-12, 13, (n, 16, 3)))
    one = rng.integers(0, 256, (n, 1, 3)) + np.zeros((n, 16, 3), np.int64)
    one[:, :, 1] += rng.integers(-60, 61, (n, 16))
    kinds.append(one)
    L = lib()
    err = np.zeros(64, np.float32)
    key = np.zeros(64, np.int32)
    checked = 0
    for tex in kinds:
        tex = np.clip(tex, 0, 255).astype(np.float32)
        blocks = np.zeros((n, 64), np.float32)
        blocks[:, :48] = tex.transpose(0, 2, 1).reshape(n, 48)
        blocks[:, 48:] = 255
        for b in range(n):
            lb = bounds_of(L, blocks[b])
            for mode in (1, 3):
                L.oracle_bc7_part_fast_errors(blocks[b].ctypes.data, mode, err.ctypes.data, key.ctypes.data)
                assert (lb <= err.astype(np.float64)).all(), (b, mode, int(np.argmax(lb - err)), float((lb - err).max()))
            checked += 1
    assert checked == 5 * n


def test_float_restatement_never_exceeds_the_bound_in_exact_arithmetic():
    """the fp32 evaluation (scaled matrix, ||M^2||_F^(1/2) for the eigenvalue, margins) against the same bound in float64 with the exact
    largest eigenvalue: the fp32 number is the smaller one for every block and shape -- roundings never push it above the mathematics"""
    import re
    rng = np.random.default_rng(99)
    t = open(os.path.join(ROOT, "oracle", "bc7_tables.h")).read()
    m = re.search(r"BCN_SUBSET_MASKS\[128\]\s*=\s*\{([^}]*)\}", t)
    masks = [int(x, 16) & 0xffff for x in re.findall(r"0x([0-9a-fA-F]+)u", m.group(1))][:64]
    n = 1500
    tex = np.concatenate([rng.integers(0, 256, (n, 16, 3)),
                          np.clip(rng.integers(0, 256, (n, 1, 3)) + rng.integers(-20, 21, (n, 16, 3)), 0, 255),
                          np.where(rng.integers(0, 2, (n, 16, 1)) == 1, rng.integers(0, 256, (n, 1, 3)), rng.integers(0, 256, (n, 1, 3)))]).astype(np.float64)
    nb = tex.shape[0]
    exact = np.zeros((nb, 64))
    for p in range(64):
        in0 = np.array([(masks[p] >> k) & 1 for k in range(16)], bool)
        for sel in (in0, ~in0):
            x = tex[:, sel, :]
            x = x - x.mean(axis=1, keepdims=True)
            c = np.einsum("bki,bkj->bij", x, x)
            r = np.maximum(np.trace(c, axis1=1, axis2=2) - np.linalg.eigvalsh(c)[:, -1], 0)
            exact[:, p] += np.maximum(np.sqrt(r) - np.sqrt(3) / 2 * np.sqrt(sel.sum()), 0) ** 2
    L = lib()
    blocks = np.zeros((nb, 64), np.float32)
    blocks[:, :48] = tex.transpose(0, 2, 1).reshape(nb, 48)
    blocks[:, 48:] = 255
    worst = 0.0
    for b in range(nb):
        lb = bounds_of(L, blocks[b])
        assert (lb <= exact[b] + 1e-9).all(), (b, int(np.argmax(lb - exact[b])), float((lb - exact[b]).max()))
        worst = max(worst, float((exact[b] - lb).max()))
    assert worst > 0


# ---- round 5: the one-line bound that lets mode 6 be skipped under an RGB profile (csrc/bc7.hip mode6_cannot_win) -------------------------

def _one_line(L):
    L.oracle_bc7_one_line_bound.argtypes = [C.c_void_p]
    L.oracle_bc7_one_line_bound.restype = C.c_float
    return L.oracle_bc7_one_line_bound


def _mode6_only():
    s = pyoracle.bc7_profile("slow")
    s.mode_selection[0] = 0; s.mode_selection[1] = 0; s.mode_selection[2] = 0; s.mode_selection[3] = 1
    return s


@pytest.mark.parametrize("name,img", list(sample_images()), ids=[n for n, _ in sample_images()])
def test_one_line_bound_never_exceeds_the_oracles_mode_6_error(name, img):
    """mode 6 alone (RGB profile, every refinement count from 0 to the slow profile's 4): its error is never below the bound"""
    L = lib()
    f = _one_line(L)
    blocks = planar_blocks(img)
    data = (C.c_uint32 * 4)()
    e = C.c_float()
    L.oracle_bc7_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_bc7_block.restype = None
    s = _mode6_only()
    tight = 0
    for it in (0, 1, 4):
        s.refineIterations[6] = it
        for b in range(blocks.shape[0]):
            lb = f(blocks[b].ctypes.data)
            L.oracle_bc7_block(blocks[b].ctypes.data, C.byref(s), data, C.byref(e))
            assert 0.0 <= lb <= e.value, (name, it, b, lb, e.value)
            tight += lb > 0.5 * e.value
    if name == "ldr_smooth":
        assert tight > 0.5 * 3 * blocks.shape[0]


def test_one_line_bound_on_random_blocks_and_against_exact_arithmetic():
    """20 000 seeded random blocks of five kinds: bound <= the oracle's mode 6 error, and the fp32 number never exceeds the same bound in
    float64 with the exact largest eigenvalue; brute-force 4-bit palettes from arbitrary integer endpoints stay above it too"""
    rng = np.random.default_rng(606)
    n = 4000
    kinds = [rng.integers(0, 256, (n, 16, 3))]
    a, b = rng.integers(0, 256, (n, 1, 3)), rng.integers(0, 256, (n, 1, 3))
    kinds.append(np.where(rng.integers(0, 2, (n, 16, 1)) == 1, a, b) + rng.integers(-3, 4, (n, 16, 3)))
    gx, gy = np.meshgrid(np.arange(4), np.arange(4))
    g = (gx.reshape(1, 16, 1) * rng.integers(-40, 41, (n, 1, 3)) + gy.reshape(1, 16, 1) * rng.integers(-40, 41, (n, 1, 3)))
    kinds.append(rng.integers(40, 216, (n, 1, 3)) + g + rng.integers(-2, 3, (n, 16, 3)))
    kinds.append(rng.integers(0, 256, (n, 1, 3)) + rng.integers(-12, 13, (n, 16, 3)))
    t = np.arange(16).reshape(1, 16, 1) * rng.integers(-17, 18, (n, 1, 3)) + rng.integers(0, 256, (n, 1, 3))      # exactly collinear (before clipping)
    kinds.append(t)
    L = lib()
    f = _one_line(L)
    data = (C.c_uint32 * 4)()
    e = C.c_float()
    L.oracle_bc7_block.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_bc7_block.restype = None
    s = _mode6_only()
    w4 = np.array([0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64])
    for tex in kinds:
        tex = np.clip(tex, 0, 255).astype(np.float64)
        x = tex - tex.mean(axis=1, keepdims=True)
        c = np.einsum("bki,bkj->bij", x, x)
        r = np.maximum(np.trace(c, axis1=1, axis2=2) - np.linalg.eigvalsh(c)[:, -1], 0)
        exact = np.maximum(np.sqrt(r) - np.sqrt(3) / 2 * 4, 0) ** 2
        blocks = np.zeros((n, 64), np.float32)
        blocks[:, :48] = tex.transpose(0, 2, 1).reshape(n, 48)
        blocks[:, 48:] = 255
        for b in range(n):
            lb = f(blocks[b].ctypes.data)
            assert lb <= exact[b] + 1e-9, (b, lb, exact[b])
            if b % 4 == 0:
                L.oracle_bc7_block(blocks[b].ctypes.data, C.byref(s), data, C.byref(e))
                assert lb <= e.value, (b, lb, e.value)
            if b % 16 == 0:          # arbitrary endpoints, best level per texel (rounded as block_quant decodes them)
                e0, e1 = rng.integers(0, 256, 3), rng.integers(0, 256, 3)
                pal = ((64 - w4)[:, None] * e0[None, :] + w4[:, None] * e1[None, :] + 32) // 64
                err = ((tex[b][:, None, :] - pal[None, :, :]) ** 2).sum(axis=2).min(axis=1).sum()
                assert lb <= err, (b, lb, err)


def test_bands_cover_every_chunk_once_and_their_list_scans_fit_their_rows():
    """Round 5 (csrc/bc7.hip: ChunkSel kind 1, launch_bc7): a `slow` / `alpha_slow` surface is cut into stripes of `stripe` chunks, even
    stripes = band 0, odd stripes = band 1; workgroup i of a band's launch takes that band's i-th chunk, the grid is whole stripes and
    workgroups behind the surface return.  Restated here: for every chunk count the two bands cover every chunk exactly once; a band's
    share of the workspace (lists, share winners: nchunks / 2 + 65 chunks) holds it; a band's list scan, whose winner rows are the band's
    chunk count long, is covered by the band's grid (list_scan_parts x listed blocks <= rows); the pilot's estimate (every eighth
    workgroup of band 0, + 3) looks at chunks of band 0 only and at 1/16 of the surface within one stripe's worth."""
    TPB, MAXP = 256, 8

    def sel(stripe, band, i):
        q = i // stripe
        return (2 * q + band) * stripe + (i - q * stripe)

    for nchunks in (32, 33, 47, 63, 64, 65, 77, 128, 1000, 1023, 1024, 1025, 4096, 65536):
        stripe = max(1, min(64, nchunks // 16))
        stripes = (nchunks + stripe - 1) // stripe
        seen = {}
        cnts = []
        for band in (0, 1):
            cnt = ((stripes + 1 - band) // 2) * stripe
            cnts.append(cnt)
            assert cnt <= nchunks // 2 + 65                      # fused_layout(): a band's region
            for i in range(cnt):
                c = sel(stripe, band, i)
                if c >= nchunks:
                    continue                                     # behind the surface: the workgroup returns
                assert c not in seen
                seen[c] = band
            # the band's list scan: rows = cnt * TPB; any list length up to every block of the band
            rows = cnt * TPB
            for count in (1, TPB, rows // 3, rows):
                chunks8 = ((count + TPB - 1) // TPB + 7) & ~7
                parts = max(1, min(MAXP, rows // (chunks8 * TPB)))
                grid = (cnt + 7) // 8 * 8                        # scan_rgb: groups x grain >= ceil8(cnt)
                assert parts * count <= rows or parts == 1
                assert ((count + TPB - 1) // TPB) * parts <= grid or parts == 1, (nchunks, band, count, parts, grid)
        assert sorted(seen) == list(range(nchunks)), nchunks
        sampled = [sel(stripe, 0, 8 * j + 3) for j in range((cnts[0] + 7) // 8)]
        sampled = [c for c in sampled if c < nchunks]
        assert all(seen[c] == 0 for c in sampled) and len(set(sampled)) == len(sampled)
        assert abs(len(sampled) - nchunks / 16) <= max(2, stripe / 8 + 1), (nchunks, len(sampled))
