"""Host-pointer BC7 calls large enough to be cut into staged runs (csrc/abi.hip compress(): >= 524 288 blocks): round 5 overlaps the runs as
deep bands on two streams and lets the first run's pilot estimate pick the launch shape of the remaining runs (deep bands, or the wide
shape one run after the other).  Both outcomes -- and the alternation between them from call to call, which exercises the cached verdict
and the workspace slices -- must emit the bytes of the device-resident call, which tests/test_gpu_parity_bc7.py pins to the oracle at
this size; a band of each result is also compared with the oracle directly."""
import os

import numpy as np
import pytest

from conftest import first_mismatch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _surfaces(golden_inputs):
    from itw_amd import surfaces
    h, w = 2048, 4096                                    # 524 288 blocks: three runs (1/8, 1/2, 3/8 of the block rows)
    noisy = surfaces.ldr_smooth(h, w)
    bab = golden_inputs["baboon"]
    nat = np.ascontiguousarray(np.tile(bab, (h // bab.shape[0], w // bab.shape[1], 1)))
    return (("noise over smooth fields: few blocks need modes 1/3 -> deep bands", noisy),
            ("photograph: nearly all do -> wide shape for the remaining runs", nat))


@pytest.mark.parametrize("prof", ["slow", "basic", "alpha_slow"])
def test_staged_runs_equal_the_device_resident_call(itw, gpu, oracle, golden_inputs, prof):
    import torch
    imgs = _surfaces(golden_inputs)
    want = {}
    for name, img in imgs:
        out = itw.compress("bc7", torch.from_numpy(img).to(gpu), prof)
        torch.cuda.synchronize()
        want[name] = out.cpu().numpy()
        band = slice(1024, 1024 + 32)                    # 32 texel rows straight against the oracle
        ref = oracle.encode_mt("bc7", np.ascontiguousarray(img[band]), prof)
        lo = (band.start // 4) * (img.shape[1] // 4) * 16
        assert first_mismatch(want[name][lo:lo + ref.size], ref, 16) is None, (name, prof)
    order = [0, 1, 1, 0, 0, 1]                           # every transition of the cached verdict
    for k in order:
        name, img = imgs[k]
        got = itw.compress_numpy("bc7", img, prof)
        assert first_mismatch(got, want[name], 16) is None, (name, prof, first_mismatch(got, want[name], 16))


def test_staged_runs_with_a_padded_pitch_and_a_device_destination(itw, gpu, golden_inputs):
    """rows further apart than their texels (the staging copy packs them), and the block stream left on the device"""
    import ctypes as C
    import torch
    name, img = _surfaces(golden_inputs)[0]
    wide = np.zeros((img.shape[0], img.shape[1] + 64, 4), np.uint8)
    wide[:, :img.shape[1]] = img
    view = wide[:, :img.shape[1]]
    out = itw.compress("bc7", torch.from_numpy(img).to(gpu), "slow")
    torch.cuda.synchronize()
    want = out.cpu().numpy()
    got = itw.compress_numpy("bc7", view, "slow")
    assert first_mismatch(got, want, 16) is None
    dst = torch.zeros(want.size, dtype=torch.uint8, device=gpu)
    surf = itw.RgbaSurface(img.ctypes.data, img.shape[1], img.shape[0], img.strides[0])
    s = itw.bc7_profile("slow")
    itw.lib().CompressBlocksBC7(C.byref(surf), C.c_void_p(dst.data_ptr()), C.byref(s))
    torch.cuda.synchronize()
    assert first_mismatch(dst.cpu().numpy(), want, 16) is None


@pytest.mark.parametrize("fmt,prof,h,w", [("bc7", "basic", 2048, 4096), ("bc7", "alpha_basic", 2048, 4096), ("bc6h", "slow", 2048, 2048)])
def test_host_pointer_windows_straight_against_the_oracle(itw, gpu, oracle, fmt, prof, h, w):
    """Round 6: a large host-pointer call of BC6H `slow` or of a BC7 profile without an order verdict runs as WINDOWS of ~131 072 blocks on two
    kernel streams + the copy stream (abi.hip compress() -> compress_sliced).  The whole result, not a band of it, against the threaded oracle --
    twice in a row (the second call reuses the staging buffers, workspace slices and events of the first), and once more with rows further
    apart than their texels."""
    from itw_amd import surfaces
    img = surfaces.hdr_smooth(h, w) if fmt == "bc6h" else surfaces.ldr_smooth(h, w)
    want = oracle.encode_mt(fmt, img, prof).reshape(-1)
    for _ in range(2):
        got = itw.compress_numpy(fmt, img, prof)
        assert first_mismatch(got, want, 16) is None, (fmt, prof, first_mismatch(got, want, 16))
    wide = np.zeros((h, w + 16, 4), img.dtype)
    wide[:, :w] = img
    got = itw.compress_numpy(fmt, wide[:, :w], prof)
    assert first_mismatch(got, want, 16) is None, (fmt, prof, "padded pitch")
