"""DDS container (include/itw_dds.h): header fields as DirectXTex's _EncodeDDSHeader writes them for the formats the
plugin saves (DirectXTexDDS.cpp:441-675, DDS.h:38-236), sizes by the pitch rule (DirectXTexUtil.cpp:601-619), data order
of SaveToDDSMemory.  Expected values below are restated from those sources field by field.  CPU only."""
import ctypes as C
import struct

import numpy as np
import pytest


def _hdr(itw, fmt_key, w, h, mips=1, cube=False, arr=1):
    d = itw.DdsDesc(w, h, mips, itw.DXGI_FORMAT[fmt_key], 1 if cube else 0, arr)
    n = itw.lib().itwDdsHeaderBytes(C.byref(d))
    buf = np.zeros(n, dtype=np.uint8)
    assert itw.lib().itwDdsWriteHeader(C.byref(d), buf.ctypes.data, n) == n
    return d, buf


def _words(buf):
    return struct.unpack("<%dI" % (buf.size // 4), buf.tobytes())


def test_legacy_fourcc_headers(itw):
    for key, cc, bpb in (("bc1", b"DXT1", 8), ("bc3", b"DXT5", 16), ("bc4", b"BC4U", 8), ("bc5", b"BC5U", 16)):   # DDS.h:71-90
        d, buf = _hdr(itw, key, 4096, 4096)
        assert buf.size == 128
        w = _words(buf)
        assert w[0] == 0x20534444                          # "DDS "
        assert w[1] == 124                                 # dwSize
        assert w[2] == 0x1007 | 0x20000 | 0x80000          # TEXTURE | MIPMAP (mipLevels > 0) | LINEARSIZE
        assert (w[3], w[4]) == (4096, 4096)                # height, width
        assert w[5] == 1024 * 1024 * bpb                   # linear size of the top level
        assert w[6] == 1 and w[7] == 1                     # depth, mip count
        assert all(v == 0 for v in w[8:19])                # reserved
        assert w[19] == 32 and w[20] == 0x4                # ddspf.dwSize, DDS_FOURCC
        assert struct.pack("<I", w[21]) == cc
        assert all(v == 0 for v in w[22:27])
        assert w[27] == 0x1000 and w[28] == 0              # caps: TEXTURE; caps2
        assert all(v == 0 for v in w[29:32])


def test_dx10_headers(itw):
    for key, fmt in (("bc7", 98), ("bc7_srgb", 99), ("bc6h", 95), ("bc1_srgb", 72), ("bc3_srgb", 78)):
        d, buf = _hdr(itw, key, 1024, 512, mips=11)
        assert buf.size == 148
        w = _words(buf)
        assert struct.pack("<I", w[21]) == b"DX10"
        assert w[7] == 11 and w[27] == (0x1000 | 0x400008)  # mip count; caps TEXTURE | COMPLEX | MIPMAP
        assert w[32:37] == (fmt, 3, 0, 1, 0)                # dxgiFormat, TEXTURE2D, miscFlag, arraySize, miscFlags2
        bpb = 8 if fmt in (71, 72) else 16
        assert w[5] == 256 * 128 * bpb


def test_cubemap_header_and_sizes(itw):
    d, buf = _hdr(itw, "bc7", 256, 256, mips=9, cube=True)
    w = _words(buf)
    assert w[27] == (0x1000 | 0x400008 | 0x8) and w[28] == 0xFE00     # COMPLEX + all six faces
    assert w[32:37] == (98, 3, 0x4, 1, 0)                              # TEXTURECUBE, one cube
    chain = sum(max(1, (256 >> m) // 4) ** 2 * 16 for m in range(9))
    assert itw.lib().itwDdsFileBytes(C.byref(d)) == 148 + 6 * chain
    # a legacy-format cube stays legacy (single cube), a texture array forces DX10 (DirectXTexDDS.cpp:450-457)
    _, b1 = _hdr(itw, "bc1", 64, 64, cube=True)
    assert b1.size == 128
    _, b2 = _hdr(itw, "bc1", 64, 64, arr=3)
    assert b2.size == 148 and _words(b2)[35] == 3


def test_level_sizes_follow_the_pitch_rule(itw):
    L = itw.lib()
    assert L.itwDdsLevelBytes(71, 4096, 4096) == 1024 * 1024 * 8
    assert L.itwDdsLevelBytes(98, 1, 1) == 16                          # max(1, (w+3)/4)
    assert L.itwDdsLevelBytes(98, 5, 9) == 2 * 3 * 16
    assert L.itwDdsLevelBytes(77, 2, 2) == 16
    assert L.itwDdsLevelBytes(0, 4, 4) == 0


def test_file_roundtrip(itw):
    rng = np.random.default_rng(11)
    w, h, mips = 64, 32, 4
    sizes = [itw.lib().itwDdsLevelBytes(98, max(1, w >> m), max(1, h >> m)) for m in range(mips)]
    levels = [rng.integers(0, 256, size=n, dtype=np.uint8) for n in sizes]
    f = itw.dds_file("bc7", w, h, levels, mip_levels=mips)
    assert f.size == 148 + sum(sizes)
    d = itw.DdsDesc()
    off = itw.lib().itwDdsReadHeader(f.ctypes.data, f.size, C.byref(d))
    assert off == 148 and (d.width, d.height, d.mip_levels, d.dxgi_format, d.is_cubemap, d.array_size) == (w, h, mips, 98, 0, 1)
    pos = off
    for lv in levels:                                                  # top level first, tightly packed
        assert np.array_equal(f[pos:pos + lv.size], lv)
        pos += lv.size
    with pytest.raises(ValueError):
        itw.dds_file("bc7", w, h, levels[:2], mip_levels=mips)         # wrong level count
    assert itw.lib().itwDdsReadHeader(f.ctypes.data, 100, C.byref(d)) == 0
    bad = f.copy(); bad[0] = 0
    assert itw.lib().itwDdsReadHeader(bad.ctypes.data, bad.size, C.byref(d)) == 0


def test_golden_blocks_as_dds(itw, golden_blocks, golden_inputs):
    """The committed golden BC1 stream of baboon.png wrapped as a DDS: 128-byte DXT1 header + the blocks."""
    key = next(k for k in golden_blocks if k.startswith("baboon") and "bc1" in k)
    blocks = golden_blocks[key]
    f = itw.dds_file("bc1", 256, 256, [blocks])
    assert f.size == 128 + blocks.size and np.array_equal(f[128:], blocks.reshape(-1))


def test_bc4_bc5_legacy_fourcc_read_both_spellings(itw):
    """DirectXTex writes 'BC4U' / 'BC5U' (DDS.h:83-90) and also accepts the older 'ATI1' / 'ATI2' (DirectXTexDDS.cpp:64-70)."""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_bc45.npz")))
    blocks = g["monkey_crop.bc5"]                                      # 217 x 215 texels -> 55 x 54 blocks
    f = itw.dds_file("bc5", 215, 217, [blocks])
    assert f.size == 128 + 54 * 55 * 16 and np.array_equal(f[128:], blocks)
    d = itw.DdsDesc()
    assert itw.lib().itwDdsReadHeader(f.ctypes.data, f.size, C.byref(d)) == 128
    assert (d.width, d.height, d.mip_levels, d.dxgi_format) == (215, 217, 1, 83)
    for cc, fmt in ((b"ATI1", 80), (b"ATI2", 83), (b"BC4U", 80)):
        alt = f.copy()
        alt[84:88] = np.frombuffer(cc, dtype=np.uint8)                 # ddspf.dwFourCC
        assert itw.lib().itwDdsReadHeader(alt.ctypes.data, alt.size, C.byref(d)) == 128 and d.dxgi_format == fmt


@pytest.mark.parametrize("key,expect,mips,cube,arr", [("bc1", "DXT1", 1, False, 1), ("bc3", "DXT5", 5, False, 1), ("bc4", "BC4U", 1, False, 1),
                                                      ("bc5", "BC5U", 3, False, 1), ("bc7", "DX10", 1, False, 1), ("bc7_srgb", "DX10", 9, True, 1),
                                                      ("bc6h", "DX10", 4, False, 3), ("bc1_srgb", "DX10", 1, False, 1)])
def test_headers_read_back_through_the_references_own_dds_definitions(itw, tmp_path, key, expect, mips, cube, arr):
    """oracle/_ref/ref_dds_check is built on DirectXTex/DDS.h compiled unmodified from the reference: struct DDS_HEADER /
    DDS_HEADER_DXT10, the DDSPF_* pixel formats and the flag macros come from there, not from this repo."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "ref_dds_check")
    if not os.path.exists(exe):
        if not os.path.exists("/root/reference/3rdParty/DirectXTex/DirectXTex/DDS.h"):
            pytest.skip("oracle/_ref/ref_dds_check not prebuilt and /root/reference absent")
        subprocess.run(["make", "-C", os.path.join(root, "oracle", "ref_build")], check=True)
    d, buf = _hdr(itw, key, 256, 128, mips=mips, cube=cube, arr=arr)
    path = tmp_path / "h.dds"
    buf.tofile(path)
    r = subprocess.run([exe, str(path), expect, str(itw.DXGI_FORMAT[key]), "256", "128", str(mips), "1" if cube else "0", str(arr)],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "OK", r.stdout + r.stderr
