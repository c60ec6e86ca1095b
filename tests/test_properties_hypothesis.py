"""Property tests of the oracle (hypothesis): the structural facts the GPU mapping and the sharding rely on -- blocks never
interact (kernel.ispc:573-596, 2014-2028, 3118-3130), ignored channels are really ignored (ispc_texcomp.h:95-103: alpha
is unused by BC1, BC6H and the RGB BC7 profiles), and any row-band split gives the same stream (win32Threads.cpp:217-231).
CPU only; the same properties hold for the HIP path through the parity tests."""
import numpy as np
from hypothesis import given, settings, strategies as st, HealthCheck

COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


def _image(draw, dtype, hmax=3, wmax=4):
    bh, bw = draw(st.integers(1, hmax)), draw(st.integers(1, wmax))
    seed = draw(st.integers(0, 2 ** 32 - 1))
    kind = draw(st.sampled_from(["noise", "flat", "two", "ramp"]))
    rng = np.random.default_rng(seed)
    hi = 256 if dtype == np.uint8 else 0x7c00          # finite positive halves for the HDR case
    if kind == "noise":
        img = rng.integers(0, hi, (bh * 4, bw * 4, 4))
    elif kind == "flat":
        img = np.broadcast_to(np.repeat(np.repeat(rng.integers(0, hi, (bh, bw, 4)), 4, 0), 4, 1), (bh * 4, bw * 4, 4))
    elif kind == "two":
        a, b = rng.integers(0, hi, (2, 4))
        img = np.where(rng.random((bh * 4, bw * 4, 1)) < 0.5, a, b)
    else:
        y, x = np.mgrid[0:bh * 4, 0:bw * 4]
        img = np.stack([(x * rng.integers(1, 9) + y * rng.integers(0, 5) + rng.integers(0, 64)) % hi for _ in range(4)], -1)
    return np.ascontiguousarray(img.astype(dtype))


ldr_images = st.composite(lambda draw: _image(draw, np.uint8))()
hdr_images = st.composite(lambda draw: _image(draw, np.uint16))()
CASES_LDR = [("bc1", None), ("bc3", None), ("bc7", "veryfast"), ("bc7", "alpha_fast"), ("bc7", "basic"), ("bc4", None), ("bc5", None)]


def _per_block(oracle, fmt, img, prof):
    h, w = img.shape[:2]
    return np.concatenate([oracle.encode(fmt, np.ascontiguousarray(img[y:y + 4, x:x + 4]), prof)
                           for y in range(0, h, 4) for x in range(0, w, 4)])


@settings(max_examples=25, **COMMON)
@given(img=ldr_images, case=st.sampled_from(CASES_LDR))
def test_blocks_never_interact_ldr(oracle, img, case):
    fmt, prof = case
    assert np.array_equal(oracle.encode(fmt, img, prof), _per_block(oracle, fmt, img, prof))


@settings(max_examples=15, **COMMON)
@given(img=hdr_images, prof=st.sampled_from(["veryfast", "basic", "slow"]))
def test_blocks_never_interact_hdr(oracle, img, prof):
    assert np.array_equal(oracle.encode("bc6h", img, prof), _per_block(oracle, "bc6h", img, prof))


@settings(max_examples=25, **COMMON)
@given(img=ldr_images, alpha_seed=st.integers(0, 2 ** 32 - 1), case=st.sampled_from([("bc1", None), ("bc7", "fast"), ("bc7", "slow"), ("bc4", None), ("bc5", None)]))
def test_alpha_is_ignored_where_the_header_says_so(oracle, img, alpha_seed, case):
    fmt, prof = case
    if prof == "slow":
        img = img[:4, :8]
    other = img.copy()
    other[..., 3] = np.random.default_rng(alpha_seed).integers(0, 256, img.shape[:2])
    if fmt in ("bc4", "bc5"):
        other[..., 2] = other[..., 3]                  # blue is unused too; BC4 also ignores green
        if fmt == "bc4":
            other[..., 1] = 255 - other[..., 3]
    assert np.array_equal(oracle.encode(fmt, np.ascontiguousarray(img), prof), oracle.encode(fmt, np.ascontiguousarray(other), prof))


@settings(max_examples=10, **COMMON)
@given(img=hdr_images, alpha_seed=st.integers(0, 2 ** 32 - 1))
def test_bc6h_ignores_alpha(oracle, img, alpha_seed):
    other = img.copy()
    other[..., 3] = np.random.default_rng(alpha_seed).integers(0, 65536, img.shape[:2])
    assert np.array_equal(oracle.encode("bc6h", img, "basic"), oracle.encode("bc6h", other, "basic"))


@settings(max_examples=20, **COMMON)
@given(seed=st.integers(0, 2 ** 32 - 1), rows=st.integers(1, 9), threads=st.integers(1, 7), case=st.sampled_from([("bc1", None), ("bc3", None), ("bc7", "veryfast")]))
def test_any_band_split_gives_the_same_stream(oracle, seed, rows, threads, case):
    fmt, prof = case
    img = np.random.default_rng(seed).integers(0, 256, (rows * 4, 16, 4), dtype=np.uint8)
    assert np.array_equal(oracle.encode_mt(fmt, img, prof, threads=threads), oracle.encode(fmt, img, prof))
