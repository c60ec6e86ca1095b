"""The reference's OWN block codecs as pins.  oracle/_ref/libdxtex_bc_ref.so = /root/reference/3rdParty/DirectXTex/DirectXTex/
{BC.cpp, BC4BC5.cpp, BC6HBC7.cpp} compiled unmodified (oracle/ref_build/Makefile; a scalar stand-in for the Windows SDK's
DirectXMath, which the reference tree does not vendor).  Two things become reference-pinned that were not:

  * every BC7 / BC6H / BC4 / BC5 stream the encoders emit decodes, under the REFERENCE's decoder (D3DXDecodeBC7, D3DXDecodeBC6HU,
    D3DXDecodeBC4U/BC5U: the plugin's preview / load path, IntelPlugin.cpp:1059, 2558), to exactly the texels of the from-spec
    decoders in oracle/ (to which the GPU decoders are bit-equal, tests/test_gpu_decode.py); BC1 / BC3 to within the reference's
    own float-vs-integer interpolation difference (<= 1 code);
  * the BC4 / BC5 ENCODER restatement (oracle/bc4_bc5.c, csrc/bc4_bc5.hip) equals D3DXEncodeBC4U / BC5U -- the function the
    plugin itself calls for these formats (IntelPlugin.cpp:271-273) -- byte for byte.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "_ref", "libdxtex_bc_ref.so")


@pytest.fixture(scope="module")
def dxtex():
    if not os.path.exists(LIB):
        if not os.path.exists("/root/reference/3rdParty/DirectXTex/DirectXTex/BC6HBC7.cpp"):
            pytest.skip("oracle/_ref/libdxtex_bc_ref.so not prebuilt and /root/reference absent")
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "ref_build")], check=True)
    L = C.CDLL(LIB)
    L.dxtex_ref_decode.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.dxtex_ref_decode.restype = C.c_int
    L.dxtex_ref_encode_bc45.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    L.dxtex_ref_encode_bc45.restype = None
    return L


def _ref_decode(L, kind, blocks, bpb):
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8).reshape(-1, bpb)
    out = np.zeros((blocks.shape[0], 16, 4), dtype=np.float32)
    for i in range(blocks.shape[0]):
        assert L.dxtex_ref_decode(kind, blocks[i].ctypes.data, out[i].ctypes.data) == 0
    return out


def _tiles(img4x4rows, w, h):
    """(H, W, C) decode -> [block][texel][C] in raster block order, texel = y*4 + x."""
    a = np.asarray(img4x4rows)
    c = a.shape[2]
    return a.reshape(h // 4, 4, w // 4, 4, c).transpose(0, 2, 1, 3, 4).reshape(-1, 16, c)


# ------------------------------------------------------------------------------------------------------------------ BC7

@pytest.mark.parametrize("image,prof", [("monkey", "slow"), ("monkey", "alpha_slow"), ("edge_cases", "alpha_basic"), ("baboon", "basic")])
def test_reference_decoder_reads_bc7_streams_like_the_from_spec_decoder(dxtex, oracle, golden_inputs, golden_blocks, image, prof):
    img = golden_inputs[image]
    h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
    blocks = golden_blocks[f"{image}.bc7.{prof}"]
    ours, modes = oracle.decode("bc7", blocks, w, h)
    ref = _ref_decode(dxtex, 7, blocks, 16)                              # floats = byte / 255
    got = np.rint(ref * 255.0).astype(np.int32)
    assert np.abs(ref * 255.0 - got).max() < 1e-3
    assert np.array_equal(got, _tiles(ours, w, h).astype(np.int32))
    if image == "monkey" and prof == "alpha_slow":
        assert set(np.unique(modes)) == set(range(8))                    # the comparison covered every BC7 mode


def test_reference_decoder_on_random_bc7_blocks(dxtex, oracle):
    """Random 128-bit words: every mode, partition, rotation, index-selector and p-bit combination (and the reserved mode,
    which the reference decodes to zero)."""
    rng = np.random.default_rng(7)
    blocks = rng.integers(0, 256, size=(4096, 16), dtype=np.uint8)
    for i in range(8):                                                   # make every mode frequent
        sel = slice(i * 400, (i + 1) * 400)
        blocks[sel, 0] = (blocks[sel, 0] & ~np.uint8((1 << (i + 1)) - 1)) | np.uint8(1 << i)
    ours, modes = oracle.decode("bc7", blocks.reshape(-1), 4 * 64, 4 * 64)
    ref = np.rint(_ref_decode(dxtex, 7, blocks, 16) * 255.0).astype(np.int32)
    mine = _tiles(ours, 256, 256).astype(np.int32)
    ok = modes >= 0
    assert ok.sum() > 3500 and np.array_equal(ref[ok], mine[ok])


# ----------------------------------------------------------------------------------------------------------------- BC6H

def _half_to_float(bits):
    return np.asarray(bits, dtype=np.uint16).view(np.float16).astype(np.float32)


@pytest.mark.parametrize("image,prof", [("monkey_hdr", "slow"), ("monkey_hdr", "fast"), ("hdr_random_bits", "slow")])
def test_reference_decoder_reads_bc6h_streams_like_the_from_spec_decoder(dxtex, oracle, golden_inputs, golden_blocks, image, prof):
    img = golden_inputs[image]
    h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
    blocks = golden_blocks[f"{image}.bc6h.{prof}"]
    ours, modes = oracle.decode("bc6h", blocks, w, h)                    # uint16 half bit patterns, RGB
    ref = _ref_decode(dxtex, 6, blocks, 16)[..., :3]
    want = _half_to_float(_tiles(ours, w, h))
    assert np.array_equal(np.nan_to_num(ref, nan=-1.0, posinf=3e38, neginf=-3e38), np.nan_to_num(want, nan=-1.0, posinf=3e38, neginf=-3e38))
    if image == "monkey_hdr" and prof == "slow":
        assert len(np.unique(modes)) >= 8


def test_reference_decoder_on_random_bc6h_blocks(dxtex, oracle):
    rng = np.random.default_rng(66)
    blocks = rng.integers(0, 256, size=(4096, 16), dtype=np.uint8)
    ours, modes = oracle.decode("bc6h", blocks.reshape(-1), 256, 256)
    ref = _ref_decode(dxtex, 6, blocks, 16)[..., :3]
    want = _half_to_float(_tiles(ours, 256, 256))
    ok = modes >= 0
    assert ok.sum() > 3000
    a = np.nan_to_num(ref[ok], nan=-1.0, posinf=3e38, neginf=-3e38)
    b = np.nan_to_num(want[ok], nan=-1.0, posinf=3e38, neginf=-3e38)
    assert np.array_equal(a, b)


# -------------------------------------------------------------------------------------------------------------- BC1 / BC3

@pytest.mark.parametrize("fmt,kind,bpb", [("bc1", 1, 8), ("bc3", 3, 16)])
def test_reference_decoder_reads_bc1_bc3_streams(dxtex, oracle, golden_inputs, golden_blocks, fmt, kind, bpb):
    """DirectXTex expands 565 endpoints as c/31, c/63 and interpolates palettes in float (BC.cpp:327-360); the from-spec decoders
    use bit replication and the integer rule.  Different definitions of the same block (neither is the plugin's encode path):
    they agree to within about one code."""
    img = golden_inputs["baboon"]
    blocks = golden_blocks[f"baboon.{fmt}"]
    ours, _ = oracle.decode(fmt, blocks, 256, 256)
    ref = _ref_decode(dxtex, kind, blocks, bpb) * 255.0
    mine = _tiles(ours, 256, 256).astype(np.float32)
    nch = 3 if fmt == "bc1" else 4
    assert np.abs(ref[..., :nch] - mine[..., :nch]).max() <= 1.5
    assert np.mean(np.abs(ref[..., :nch] - mine[..., :nch])) < 0.4


# -------------------------------------------------------------------------------------------------------------- BC4 / BC5

def _bc45_blocks(n, seed):
    """Channel blocks as DirectXTex's loader produces them: byte / 255 in fp32 (code x (1/255.f), DESIGN r01 7.6)."""
    rng = np.random.default_rng(seed)
    kinds = []
    kinds.append(rng.integers(0, 256, size=(n, 16)))                                           # noise
    base = rng.integers(0, 256, size=(n, 1)); kinds.append(np.clip(base + rng.integers(-6, 7, size=(n, 16)), 0, 255))   # near-flat
    kinds.append(np.repeat(rng.integers(0, 256, size=(n, 1)), 16, axis=1))                     # flat
    two = rng.integers(0, 256, size=(n, 2)); kinds.append(two[np.arange(n)[:, None], rng.integers(0, 2, size=(n, 16))])   # two-level
    ramp = np.clip(rng.integers(0, 200, size=(n, 1)) + np.arange(16)[None, :] * rng.integers(0, 5, size=(n, 1)), 0, 255); kinds.append(ramp)
    edge = rng.choice([0, 255, 1, 254, 128], size=(n, 16)); kinds.append(edge)                  # boundary-heavy: drives the 6-step codec
    codes = np.concatenate(kinds, axis=0).astype(np.uint8)
    return codes, (codes.astype(np.float32) * np.float32(1.0 / 255.0)).astype(np.float32)


def test_bc4_encoder_restatement_equals_the_reference_encoder(dxtex, oracle):
    codes, tex = _bc45_blocks(400, 45)
    for i in range(tex.shape[0]):
        rg = np.zeros((16, 2), dtype=np.float32)
        rg[:, 0] = tex[i]
        ref = np.zeros(8, dtype=np.uint8)
        dxtex.dxtex_ref_encode_bc45(1, rg.ctypes.data, ref.ctypes.data)
        mine = oracle.bc4_block(tex[i])
        assert np.array_equal(ref, mine), (i, codes[i].tolist(), ref.tobytes().hex(), mine.tobytes().hex())


def test_bc5_surface_restatement_equals_the_reference_encoder_block_by_block(dxtex, oracle):
    """Whole path of the oracle (surface walk, R and G of RGBA8) against the reference encoder fed the same texels."""
    from itw_amd import surfaces
    img = np.ascontiguousarray(np.concatenate([surfaces.ldr_smooth(32, 64), surfaces.ldr_uniform(32, 64)], axis=0))
    mine = oracle.encode("bc5", img).reshape(-1, 16)
    t = _tiles(img, 64, 64).astype(np.float32) * np.float32(1.0 / 255.0)
    for i in range(t.shape[0]):
        rg = np.ascontiguousarray(t[i, :, :2], dtype=np.float32)
        ref = np.zeros(16, dtype=np.uint8)
        dxtex.dxtex_ref_encode_bc45(2, rg.ctypes.data, ref.ctypes.data)
        assert np.array_equal(ref, mine[i]), (i, ref.tobytes().hex(), mine[i].tobytes().hex())


def test_bc4_float_decode_restatement_equals_the_reference_decoder(dxtex, oracle):
    rng = np.random.default_rng(4)
    blocks = rng.integers(0, 256, size=(512, 8), dtype=np.uint8)
    blocks[:64, 0] = blocks[:64, 1]                                       # red_0 == red_1
    ref = _ref_decode(dxtex, 4, blocks, 8)[..., 0]
    L = oracle.lib()
    for i in range(blocks.shape[0]):
        out = np.zeros(16, dtype=np.float32)
        L.oracle_decode_bc4_float(blocks[i].ctypes.data, out.ctypes.data)
        assert np.array_equal(out, ref[i]), i


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,nch", [("bc4", 1), ("bc5", 2)])
def test_gpu_bc45_encoder_equals_the_reference_encoder(dxtex, itw, gpu, fmt, nch):
    """csrc/bc4_bc5.hip against D3DXEncodeBC4U / BC5U compiled from the reference, block by block (no oracle in between)."""
    from itw_amd import surfaces
    img = np.ascontiguousarray(np.concatenate([surfaces.ldr_smooth(64, 128), surfaces.ldr_uniform(32, 128), surfaces.ldr_edge_cases()[:32].repeat(2, axis=1)], axis=0))
    h, w = img.shape[:2]
    got = itw.compress_numpy(fmt, img).reshape(-1, 8 * nch)
    t = _tiles(img, w, h).astype(np.float32) * np.float32(1.0 / 255.0)
    for i in range(t.shape[0]):
        rg = np.ascontiguousarray(t[i, :, :2], dtype=np.float32)
        ref = np.zeros(8 * nch, dtype=np.uint8)
        dxtex.dxtex_ref_encode_bc45(nch, rg.ctypes.data, ref.ctypes.data)
        assert np.array_equal(ref, got[i]), (i, ref.tobytes().hex(), got[i].tobytes().hex())


@pytest.mark.gpu
def test_gpu_decoders_equal_the_reference_decoders(dxtex, itw, gpu, golden_inputs, golden_blocks):
    """csrc/decode.hip against D3DXDecodeBC7 / BC6HU compiled from the reference, directly."""
    for image, fmt, prof, kind in (("monkey", "bc7", "alpha_slow", 7), ("monkey_hdr", "bc6h", "slow", 6)):
        img = golden_inputs[image]
        h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
        blocks = golden_blocks[f"{image}.{fmt}.{prof}"]
        dec = itw.decode(fmt, np.ascontiguousarray(blocks), w, h)
        ref = _ref_decode(dxtex, kind, blocks, 16)
        if fmt == "bc7":
            assert np.array_equal(np.rint(ref * 255.0).astype(np.int32), _tiles(dec, w, h).astype(np.int32))
        else:
            want = _half_to_float(_tiles(dec[..., :3], w, h))
            assert np.array_equal(np.nan_to_num(ref[..., :3], nan=-1.0, posinf=3e38, neginf=-3e38), np.nan_to_num(want, nan=-1.0, posinf=3e38, neginf=-3e38))
