"""Minimal Radiance .hdr (32-bit_rle_rgbe) reader -> float32 RGB.  Needed because the reference's HDR
sample images are RGBE (Sample Images/monkey-32bit.hdr, HDR.hdr) and PIL cannot read them."""
import numpy as np


def read_hdr(path):
    with open(path, "rb") as f:
        data = f.read()
    pos = 0
    header_end = data.index(b"\n\n") + 2
    header = data[:header_end].decode("latin1")
    assert header.startswith("#?"), "not a Radiance file"
    assert "32-bit_rle_rgbe" in header
    pos = header_end
    line_end = data.index(b"\n", pos)
    dims = data[pos:line_end].decode().split()
    pos = line_end + 1
    assert dims[0] == "-Y" and dims[2] == "+X", dims
    h, w = int(dims[1]), int(dims[3])
    rgbe = np.zeros((h, w, 4), dtype=np.uint8)
    for y in range(h):
        if w < 8 or w > 0x7FFF or data[pos] != 2 or data[pos + 1] != 2 or (data[pos + 2] & 0x80):
            # flat (uncompressed) scanline
            rgbe[y] = np.frombuffer(data, dtype=np.uint8, count=w * 4, offset=pos).reshape(w, 4)
            pos += w * 4
            continue
        assert ((data[pos + 2] << 8) | data[pos + 3]) == w
        pos += 4
        for c in range(4):
            x = 0
            while x < w:
                n = data[pos]; pos += 1
                if n > 128:
                    n -= 128
                    rgbe[y, x:x + n, c] = data[pos]; pos += 1
                else:
                    rgbe[y, x:x + n, c] = np.frombuffer(data, dtype=np.uint8, count=n, offset=pos); pos += n
                x += n
    e = rgbe[..., 3].astype(np.int32)
    scale = np.where(e > 0, np.ldexp(np.float32(1.0), e - 136), np.float32(0.0)).astype(np.float32)
    return (rgbe[..., :3].astype(np.float32) + np.float32(0.5)) * scale[..., None] * (e > 0)[..., None]
