import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "intel-texture-works-plugin_amd"))
import numpy as np, torch, itw_amd
from itw_amd import surfaces
dev = torch.device("cuda:0")
base = torch.from_numpy(surfaces.ldr_smooth(4096, 4096)).to(dev)
big = base.repeat(4, 4, 1).contiguous()                     # 16384^2: offsets up to 2^30
for fmt in ("bc4", "bc5"):
    bpb = itw_amd.BYTES_PER_BLOCK[fmt]
    small = itw_amd.compress(fmt, base).view(1024, 1024, bpb)
    out = itw_amd.compress(fmt, big).view(4096, 4096, bpb)
    torch.cuda.synchronize()
    ok = all(torch.equal(out[1024 * i:1024 * (i + 1), 1024 * j:1024 * (j + 1)], small) for i in range(4) for j in range(4))
    # a strided view (stride larger than the row) and an odd row count of blocks
    sub = big[8:8 + 4 * 777, 16:16 + 4 * 333]
    o2 = itw_amd.compress(fmt, sub).view(777, 333, bpb)
    o3 = itw_amd.compress(fmt, sub.contiguous()).view(777, 333, bpb)
    torch.cuda.synchronize()
    print(fmt, "16384^2 tiles equal the 4096^2 encode:", ok, "| strided view equals its contiguous copy:", torch.equal(o2, o3))
