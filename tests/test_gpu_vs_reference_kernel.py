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
    """The prebuilt reference-source library MUST have travelled to the GPU box (a silent skip here turned
    "parity vs the reference source" into "parity vs the oracle" with a green result).  ITW_ALLOW_NO_REF=1 downgrades the
    failure to a skip for a deliberate run without it; test_every_pinned_stream below needs no binary either way."""
    import os
    from oracle import pyref            # checker only
    if not pyref.available():
        if os.environ.get("ITW_ALLOW_NO_REF") == "1":
            pytest.skip("oracle/_ref/libispc_texcomp_ref_full.so absent and ITW_ALLOW_NO_REF=1")
        pytest.fail("oracle/_ref/libispc_texcomp_ref_full.so did not travel to this box (build it in the container: "
                    "make -C oracle/ref_build; or set ITW_ALLOW_NO_REF=1 to run without the reference-source check)")
    return pyref


def test_every_pinned_stream(itw, gpu, golden_inputs):
    """HIP kernels against tests/golden/ref_full_sha256.json -- the SHA-256 of the reference SOURCE's output for every golden
    input x format x preset, generated in the container (tools/make_ref_full_sha256.py).  Needs neither /root/reference nor
    oracle/_ref at run time; both BC7 launch shapes."""
    import hashlib
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "tests", "golden", "ref_full_sha256.json")) as f:
        pins = json.load(f)["streams"]
    assert len(pins) == 46
    for key, want in pins.items():
        parts = key.split(".")
        name, fmt, prof = parts[0], parts[1], (parts[2] if len(parts) > 2 else None)
        for path in (("deep", "wide") if fmt == "bc7" else ("auto",)):
            itw.set_bc7_path(path)
            try:
                got = _gpu(itw, gpu, fmt, golden_inputs[name], prof)
            finally:
                itw.set_bc7_path("auto")
            assert hashlib.sha256(got.tobytes()).hexdigest() == want, (key, path)


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
