"""Photoshop-buffer -> encoder-surface conversions (IntelPlugin.cpp:741-810, :291-366): the oracle's restatement against
independent numpy formulas (CPU), and the device kernels of csrc/convert.hip against the oracle (GPU).  Exact for every
path -- since round 6 also 32-bit -> 8-bit with gamma: the kernel counts the code thresholds of the reference's own function
(csrc/gamma_thresholds.h) instead of calling a device pow()."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle8(oracle, src, depth, planes, has_alpha, gamma, w, h):
    out = np.zeros((h, w, 4), dtype=np.uint8)
    oracle.lib().oracle_convert_rgba8(src.ctypes.data_as(C.c_void_p), depth, planes, has_alpha, gamma, w, h, out.ctypes.data_as(C.c_void_p))
    return out


def _oracle16(oracle, src, depth, planes, has_alpha, w, h):
    out = np.zeros((h, w, 4), dtype=np.uint16)
    oracle.lib().oracle_convert_rgba16f(src.ctypes.data_as(C.c_void_p), depth, planes, has_alpha, w, h, out.ctypes.data_as(C.c_void_p))
    return out


def _source(rng, depth, planes, w, h):
    if depth == 8:
        return rng.integers(0, 256, size=(h, w, planes), dtype=np.uint8)
    if depth == 16:
        s = rng.integers(0, 32769, size=(h, w, planes)).astype(np.uint16)       # Photoshop's 0..32768
        s.reshape(-1)[:6] = [0, 1, 32767, 32768, 40000, 65535]
        return s
    s = rng.random((h, w, planes), dtype=np.float32) * np.float32(1.3) - np.float32(0.1)   # some < 0 and > 1
    s.reshape(-1)[:6] = [0.0, 1.0, 0.5, 1e-8, 0.999999, 65504.0]
    return s


def test_oracle_conversions_match_the_formulas(oracle):
    rng = np.random.default_rng(4)
    w, h = 37, 11
    for planes, alpha in ((1, 0), (3, 0), (4, 1), (4, 0)):
        s8 = _source(rng, 8, planes, w, h)
        got = _oracle8(oracle, s8, 8, planes, alpha, 0, w, h)
        want = np.zeros((h, w, 4), np.uint8); want[..., 3] = 255
        want[..., :min(planes, 3)] = s8[..., :min(planes, 3)]
        if alpha: want[..., 3] = s8[..., 3]
        assert np.array_equal(got, want)
        s16 = _source(rng, 16, planes, w, h)
        got = _oracle8(oracle, s16, 16, planes, alpha, 0, w, h)
        conv = np.where(s16 > 32768, 255, (s16.astype(np.int64) * 255) >> 15).astype(np.uint8)
        want = np.zeros((h, w, 4), np.uint8); want[..., 3] = 255
        want[..., :min(planes, 3)] = conv[..., :min(planes, 3)]
        if alpha: want[..., 3] = conv[..., 3]
        assert np.array_equal(got, want)
        got16 = _oracle16(oracle, s8, 8, planes, alpha, w, h)
        conv = (s8.astype(np.float32) / np.float32(255)).astype(np.float16).view(np.uint16)
        want = np.zeros((h, w, 4), np.uint16); want[..., 3] = 0x3C00
        want[..., :min(planes, 3)] = conv[..., :min(planes, 3)]
        if alpha: want[..., 3] = conv[..., 3]
        assert np.array_equal(got16, want)
        s32 = _source(rng, 32, planes, w, h)
        got16 = _oracle16(oracle, s32, 32, planes, alpha, w, h)
        conv = s32.astype(np.float16).view(np.uint16)                 # IEEE round to nearest even = XMConvertFloatToHalf in range
        want = np.zeros((h, w, 4), np.uint16); want[..., 3] = 0x3C00
        want[..., :min(planes, 3)] = conv[..., :min(planes, 3)]
        if alpha: want[..., 3] = conv[..., 2]                         # the reference reads plane 2 (IntelPlugin.cpp:361)
        assert np.array_equal(got16, want)


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [8, 16, 32])
def test_device_conversions_match_the_oracle(itw, gpu, oracle, depth):
    import torch
    rng = np.random.default_rng(depth)
    w, h = 203, 57
    L = itw.lib()
    L.itwSetStream(torch.cuda.current_stream().cuda_stream)
    for planes, alpha in ((1, 0), (2, 0), (3, 0), (4, 1), (4, 0)):
        src = _source(rng, depth, planes, w, h)
        d_src = torch.from_numpy(src.view(np.int16) if depth == 16 else src).to(gpu)
        for gamma in ((0, 1) if depth == 32 else (0,)):
            d8 = torch.zeros((h, w, 4), dtype=torch.uint8, device=gpu)
            assert L.itwConvertToRGBA8Device(d_src.data_ptr(), depth, planes, alpha, gamma, w, h, d8.data_ptr()) == 0
            torch.cuda.synchronize()
            got, want = d8.cpu().numpy(), _oracle8(oracle, src, depth, planes, alpha, gamma, w, h)
            assert np.array_equal(got, want), (depth, planes, alpha, gamma)       # the gamma route too: thresholds, not a device pow
        d16 = torch.zeros((h, w, 4), dtype=torch.int16, device=gpu)
        assert L.itwConvertToRGBA16FDevice(d_src.data_ptr(), depth, planes, alpha, w, h, d16.data_ptr()) == 0
        torch.cuda.synchronize()
        assert np.array_equal(d16.cpu().numpy().view(np.uint16), _oracle16(oracle, src, depth, planes, alpha, w, h))
    assert L.itwConvertToRGBA8Device(d_src.data_ptr(), 12, 3, 0, 0, w, h, d8.data_ptr()) == -1
    assert L.itwConvertToRGBA8Device(d_src.data_ptr(), depth, 3, 1, 0, w, h, d8.data_ptr()) == -1   # alpha needs plane 3


def _thresholds():
    import re
    text = open(os.path.join(ROOT, "intel-texture-works-plugin_amd", "csrc", "gamma_thresholds.h")).read()
    return np.array([int(x, 16) for x in re.findall(r"0x([0-9a-f]{8})u", text)], dtype=np.uint32)


def test_gamma_thresholds_are_the_code_boundaries_of_the_oracle(oracle):
    """csrc/gamma_thresholds.h (generated from the reference's ConvertTo8Bit compiled here): entry c is the smallest float whose code is >= c --
    checked against oracle/prepass.c (the same C library pow) at every threshold and its neighbours, increasing, and [255] <= 1.0."""
    thr = _thresholds()
    assert thr.size == 256 and thr[0] == 0 and np.all(np.diff(thr[1:].astype(np.int64)) > 0) and thr[255] <= np.float32(1.0).view(np.uint32)
    bits = (thr[1:, None].astype(np.int64) + np.arange(-2, 3)[None, :]).astype(np.uint32)          # 255 x 5 floats around the crossings
    src = bits.view(np.float32).reshape(-1)
    got = _oracle8(oracle, np.ascontiguousarray(src), 32, 1, 0, 1, src.size, 1)[..., 0].reshape(255, 5)
    c = np.arange(1, 256)[:, None]
    assert np.array_equal(got >= c, np.broadcast_to(np.arange(-2, 3)[None, :] >= 0, (255, 5)))


@pytest.mark.gpu
def test_gamma_route_is_bit_exact_at_every_code_boundary_and_for_special_values(itw, gpu, oracle):
    """The 32-bit -> 8-bit gamma route (IntelPlugin.h:66-73) on the device equals the C library route bit for bit: every threshold +-3 floats,
    two million random floats of [0, 1.2), zeros of both signs, denormals, negatives, values above 1, infinities and NaN."""
    import torch
    thr = _thresholds()
    rng = np.random.default_rng(32)
    edge = (thr[1:, None].astype(np.int64) + np.arange(-3, 4)[None, :]).astype(np.uint32).view(np.float32).reshape(-1)
    special = np.array([0.0, -0.0, 1e-45, 1e-40, -1e-40, -1.0, -np.inf, 1.0, 1.0000001, 2.0, 1e30, np.inf, np.nan, 0.5, 0.2176, 0.9999999], dtype=np.float32)
    src = np.concatenate([edge, special, rng.random(2_000_000, dtype=np.float32) * np.float32(1.2),
                          np.exp(rng.uniform(-40, 0.1, 200_000)).astype(np.float32)])
    src = np.ascontiguousarray(src[:src.size // 4 * 4])
    n = src.size
    d_src = torch.from_numpy(src).to(gpu)
    d8 = torch.zeros((1, n, 4), dtype=torch.uint8, device=gpu)
    itw.lib().itwSetStream(torch.cuda.current_stream().cuda_stream)
    assert itw.lib().itwConvertToRGBA8Device(d_src.data_ptr(), 32, 1, 0, 1, n, 1, d8.data_ptr()) == 0
    torch.cuda.synchronize()
    got = d8.cpu().numpy()[0, :, 0]
    want = _oracle8(oracle, src, 32, 1, 0, 1, n, 1)[0, :, 0]
    bad = np.nonzero(got != want)[0]
    assert bad.size == 0, (bad[:5], src[bad[:5]], got[bad[:5]], want[bad[:5]])
