"""itwCompressImageMultiGPU (include/itw_multigpu.h, csrc/multigpu.hip): one surface over R ranks from ONE process, host
code in C++ -- band per rank (win32Threads.cpp:217-231 on block rows), scatter of a device-resident surface, gather of the
output bands to the owner of `output`, each rank's first half-band gathered under the encode of its second.  The GPU box has
one device, so R ranks share it (rank r -> device r % 1): every code path except the RCCL send/recv itself runs -- band
geometry, half-band events, in-place encode on the owner, peer-copy gather, host upload / download per rank."""
import numpy as np
import pytest

from conftest import first_mismatch

pytestmark = pytest.mark.gpu


def _img(fmt, h, w):
    from itw_amd import surfaces
    return surfaces.hdr_smooth(h, w) if fmt == "bc6h" else surfaces.ldr_smooth(h, w)


@pytest.mark.parametrize("fmt,prof,h,w", [("bc1", None, 256, 256), ("bc7", "basic", 200, 128), ("bc7", "alpha_slow", 64, 64),
                                          ("bc6h", "fast", 128, 64), ("bc5", None, 61, 70), ("bc3", None, 12, 64)])
@pytest.mark.parametrize("ranks", [1, 3, 8])
def test_host_to_host(itw, gpu, oracle, fmt, prof, h, w, ranks):
    img = _img(fmt, h, w)
    want = oracle.encode(fmt, img, prof).reshape(-1)
    got = itw.compress_image_multigpu(fmt, img, prof, ranks=ranks)
    bpb = itw.BYTES_PER_BLOCK[fmt]
    assert first_mismatch(got, want, bpb) is None, first_mismatch(got, want, bpb)


@pytest.mark.parametrize("fmt,prof", [("bc1", None), ("bc7", "slow"), ("bc6h", "slow")])
def test_device_resident_surface_and_output(itw, gpu, oracle, fmt, prof):
    """Texels resident on a GPU, block stream wanted on a GPU: the owner's ranks encode in place, nothing touches the host."""
    import torch
    img = _img(fmt, 128, 192)
    want = oracle.encode_mt(fmt, img, prof).reshape(-1)
    d = torch.from_numpy(img).to(gpu)
    for ranks in (2, 8):
        out = itw.compress_image_multigpu(fmt, d, prof, ranks=ranks)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert first_mismatch(got, want, itw.BYTES_PER_BLOCK[fmt]) is None, (ranks, first_mismatch(got, want, itw.BYTES_PER_BLOCK[fmt]))
    assert itw.lib().itwMultiGpuTransport() in (b"peer", b"rccl")


def test_mixed_placements(itw, gpu, oracle):
    import torch
    img = _img("bc7", 96, 128)
    want = oracle.encode("bc7", img, "veryfast").reshape(-1)
    d = torch.from_numpy(img).to(gpu)
    host_out = itw.compress_image_multigpu("bc7", d, "veryfast", ranks=4, out=np.empty(want.size, dtype=np.uint8))     # device -> host
    assert first_mismatch(host_out, want, 16) is None
    dev_out = itw.compress_image_multigpu("bc7", img, "veryfast", ranks=4, out=torch.empty(want.size, dtype=torch.uint8, device=gpu))   # host -> device
    torch.cuda.synchronize()
    assert first_mismatch(dev_out.cpu().numpy(), want, 16) is None


def test_more_ranks_than_block_rows_and_default_rank_count(itw, gpu, oracle):
    img = _img("bc1", 8, 64)                                    # two block rows
    want = oracle.encode("bc1", img).reshape(-1)
    got = itw.compress_image_multigpu("bc1", img, ranks=16)
    assert first_mismatch(got, want, 8) is None
    assert itw.lib().itwMultiGpuRanks() >= 1
    got = itw.compress_image_multigpu("bc1", img)               # ranks = 0: one per visible device
    assert first_mismatch(got, want, 8) is None
