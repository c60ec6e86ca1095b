"""Device-pointer calls are plain asynchronous work on the caller's stream (include/itw_amd.h), so a host can capture them into a hipGraph
and replay it -- including the BC7 calls that fork to the library's second stream and join back (two bands, the pilot's gated launches,
the wide shape's side stream): the fork / join events carry the capture across.  The one call that cannot be captured is the first BC4 /
BC5 call on a device, which allocates its index table: itwWarmupBC45() does that ahead of time.  Replays must reproduce the
eager call's bytes (which the parity suites pin to the oracle), also when replayed back to back."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fmt,prof,h,w", [("bc1", None, 256, 256), ("bc3", None, 64, 512), ("bc4", None, 61, 70), ("bc5", None, 256, 256),
                                          ("bc7", "slow", 256, 256),            # wide shape: scans + side stream
                                          ("bc7", "slow", 2048, 1024),          # fused shape: two bands, pilot, gated continuations, list scans
                                          ("bc7", "alpha_slow", 1024, 1024),    # the seven-launch chain in two bands
                                          ("bc7", "basic", 1024, 2048), ("bc6h", "slow", 128, 128), ("bc6h", "fast", 512, 1024)])
def test_a_captured_call_replays_to_the_eager_bytes(itw, gpu, fmt, prof, h, w):
    import torch
    from itw_amd import surfaces
    itw.lib().itwWarmupBC45()
    img = surfaces.hdr_smooth(h, w) if fmt == "bc6h" else surfaces.ldr_smooth(h, w)
    d = torch.from_numpy(img).to(gpu)
    want = itw.compress(fmt, d, prof)
    torch.cuda.synchronize()
    want = want.clone()
    out = torch.zeros_like(want)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):                       # sizes the per-thread workspace outside the capture
        itw.compress(fmt, d, prof, out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        itw.compress(fmt, d, prof, out=out)
    torch.cuda.synchronize()
    for replays in (1, 3):
        out.zero_()
        for _ in range(replays):
            g.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want), (fmt, prof, replays)
    assert itw.last_error() is None
