"""GPU: the device implementation of the pinned arithmetic (csrc/x86_math.hpp) is bit-identical to the oracle's
(oracle/x86_math.h) -- packed 16-bit seed tables, Newton steps without FMA contraction, cvttps2dq emulation."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs():
    rng = np.random.default_rng(7)
    specials = np.array([0x00000000, 0x80000000, 0x00000001, 0x007FFFFF, 0x00800000, 0x3F800000, 0xBF800000,
                         0x7F7FFFFF, 0x7F800000, 0xFF800000, 0x7FC00000, 0x7F800001, 0xFFC00000, 0x7F000000,
                         0x7E800000, 0x4F000000, 0xCF000000, 0xCF000001, 0x4EFFFFFF], dtype=np.uint32)
    rnd = rng.integers(0, 1 << 32, size=1 << 22, dtype=np.uint64).astype(np.uint32)
    # dense mantissa sweep at the magnitudes the encoders produce (counts 1..16, norms, determinants)
    sweep = (np.float32(1.0) + np.arange(1 << 20, dtype=np.float32) * np.float32(2.0 ** -20)).view(np.uint32)
    small = np.arange(1, 4097, dtype=np.float32).view(np.uint32)
    return np.concatenate([specials, rnd, sweep, sweep + np.uint32(1 << 23), small])


def _oracle_map(oracle, fn, xs, restype):
    import ctypes as C
    f = getattr(oracle.lib(), fn)
    out = np.empty(xs.size, dtype=restype)
    xf = xs.view(np.float32)
    for i in range(xs.size):
        out[i] = f(float(xf[i])) if restype != np.uint32 else 0
    return out


def test_device_rcp_rsqrt_f2i_bit_exact(itw, gpu, oracle):
    import torch
    xs = _inputs()
    # the ctypes loop is slow; check all specials + a 200k subsample against the oracle, everything for NaN-consistency
    idx = np.concatenate([np.arange(19), np.random.default_rng(1).choice(xs.size, 200000, replace=False)])
    sub = np.ascontiguousarray(xs[idx])
    d_in = torch.from_numpy(sub.view(np.int32)).to(gpu).view(torch.float32)
    d_out = torch.empty_like(d_in)
    L = itw.test_lib()             # the self-test kernels live in the hooks build only (include/itw_test_hooks.h)
    L.itwSetStream(torch.cuda.current_stream().cuda_stream)
    O = oracle.lib()
    xf = sub.view(np.float32)
    for name, ofn in (("itwTestRcp", O.oracle_rcp), ("itwTestRsqrt", O.oracle_rsqrt)):
        getattr(L, name)(d_in.data_ptr(), d_out.data_ptr(), d_in.numel())
        torch.cuda.synchronize()
        got = d_out.cpu().numpy().view(np.uint32)
        want = np.array([ofn(float(v)) for v in xf], dtype=np.float32).view(np.uint32)
        # NaN payloads: x86 keeps the operand's payload through mulps/subps with a constant; so does the oracle's
        # C arithmetic; GPU VALU quiets and keeps payload as well -- require exact equality except NaN-vs-NaN
        nan = np.isnan(got.view(np.float32)) & np.isnan(want.view(np.float32))
        bad = (got != want) & ~nan
        assert not bad.any(), (name, hex(int(sub[bad][0])), hex(int(got[bad][0])), hex(int(want[bad][0])))
    d_i = torch.empty(d_in.numel(), dtype=torch.int32, device=gpu)
    L.itwTestF2I(d_in.data_ptr(), d_i.data_ptr(), d_in.numel())
    torch.cuda.synchronize()
    got = d_i.cpu().numpy()
    want = np.array([O.oracle_f2i(float(v)) for v in xf], dtype=np.int32)
    assert (got == want).all()
