"""CPU campaign: oracle/bc4_bc5.c against the reference's own D3DXEncodeBC4U (DirectXTex BC4BC5.cpp compiled unmodified into
oracle/_ref/libdxtex_bc_ref.so) on six classes of channel blocks (noise, near-flat, flat, two-level, ramps, boundary codes).
Usage: python tools/reference_bc45_campaign.py [blocks_per_class]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np
import test_reference_codecs as t
from oracle import pyoracle

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdxtex_bc_ref.so"))
L.dxtex_ref_encode_bc45.argtypes = [C.c_int, C.c_void_p, C.c_void_p]; L.dxtex_ref_encode_bc45.restype = None
codes, tex = t._bc45_blocks(n, 4545)
rg, ref, bad = np.zeros((16, 2), dtype=np.float32), np.zeros(8, dtype=np.uint8), 0
for i in range(tex.shape[0]):
    rg[:, 0] = tex[i]
    L.dxtex_ref_encode_bc45(1, rg.ctypes.data, ref.ctypes.data)
    bad += not np.array_equal(ref, pyoracle.bc4_block(tex[i]))
print(f"{tex.shape[0]} BC4 channel blocks (6 classes x {n}): mismatches vs the reference's D3DXEncodeBC4U: {bad}")
sys.exit(1 if bad else 0)
