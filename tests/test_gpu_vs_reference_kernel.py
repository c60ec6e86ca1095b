"""HIP kernels against the reference's own kernel.ispc, with no restatement in between: oracle/_ref/
libispc_texcomp_ref_full.so is the reference library built without ispc (kernel.ispc compiled as one scalar program
instance + the unmodified ispc_texcomp.cpp; oracle/ref_build/ispc_as_cpp/, tests/test_reference_kernel_source.py).  The
prebuilt library travels to the GPU box; /root/reference is not needed at run time."""
import numpy as np
import pytest

from conftest import first_mismatch

pytestmark = pytest.mark.gpu

BC7 = ["ultrafast", "veryfast", "fast", "basic", "slow",
       "alpha_ultrafast", "alpha_veryfast", "alpha_fast", "alpha_basic", "alpha_slow"]
BC6H = ["veryfast", "fast", "basic", "slow", "veryslow"]


@pytest.fixture(scope="module")
def refk():
    from oracle import pyref            # checker only
    if not pyref.available():
        pytest.skip("oracle/_ref/libispc_texcomp_ref_full.so not built")
    return pyref


def _gpu(itw, gpu, fmt, img, prof):
    import torch
    t = torch.from_numpy(img.view(np.int16) if fmt == "bc6h" else img).to(gpu)
    out = itw.compress(fmt, t, prof)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.parametrize("name", ["monkey", "edge_cases"])
def test_ldr_every_preset(itw, gpu, refk, golden_inputs, name):
    img = golden_inputs[name]
    for fmt, prof in [("bc1", None), ("bc3", None)] + [("bc7", p) for p in BC7]:
        want = refk.encode_mt(fmt, img, prof)
        for path in (("deep", "wide") if fmt == "bc7" else ("auto",)):
            itw.set_bc7_path(path)
            try:
                got = _gpu(itw, gpu, fmt, img, prof)
            finally:
                itw.set_bc7_path("auto")
            bpb = 8 if fmt == "bc1" else 16
            assert first_mismatch(got, want, bpb) is None, (fmt, prof, path, first_mismatch(got, want, bpb))


@pytest.mark.parametrize("name", ["monkey_hdr", "hdr_random_bits"])
def test_hdr_every_preset(itw, gpu, refk, golden_inputs, name):
    img = golden_inputs[name]
    for prof in BC6H:
        want = refk.encode_mt("bc6h", img, prof)
        got = _gpu(itw, gpu, "bc6h", img, prof)
        assert first_mismatch(got, want, 16) is None, (prof, first_mismatch(got, want, 16))


def test_bench_surface_sample_and_alpha_waves(itw, gpu, refk):
    """A 512 x 1024 cut of the bench surface (32 768 blocks: whole waves of the fused launch shape) with opaque, nearly opaque
    and translucent alpha columns, `slow` and `alpha_slow`."""
    from itw_amd import surfaces
    rng = np.random.default_rng(3)
    img = surfaces.ldr_smooth(4096, 4096)[1024:1536, 2048:3072].copy()
    img[:, 256:512, 3] = 255
    img[:, 512:768, 3] = np.where(rng.random((512, 256)) < 0.05, 254, 255)
    for prof in ("slow", "alpha_slow", "alpha_basic"):
        want = refk.encode_mt("bc7", img, prof)
        itw.set_bc7_path("deep")
        try:
            got = _gpu(itw, gpu, "bc7", img, prof)
        finally:
            itw.set_bc7_path("auto")
        assert first_mismatch(got, want, 16) is None, (prof, first_mismatch(got, want, 16))
