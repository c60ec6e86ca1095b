"""Large GPU-vs-oracle parity campaign (run on the GPU box): every format / profile on several megapixels of mixed
content -- smooth + noise, uniform random bytes, posterised (tie-heavy), real alpha, adversarial half bits -- compared
bit for bit with the multi-threaded scalar oracle.  Prints one line per case and a summary; exit code 1 on any mismatch.
Usage: python tools/parity_campaign.py [megapixels_per_case] [oracle|ref] [fmt,fmt,...] [profile,...]   (default 2, oracle, every format; the oracle needs ~1 s per
Mpix of BC7 slow on 16 cores).  `ref`: the checker is the reference's own kernel.ispc built as a scalar program
(oracle/_ref/libispc_texcomp_ref_full.so) instead of the oracle's restatement; BC4/BC5, which kernel.ispc does not have, stay on the oracle.
CAMPAIGN_SEED=<n> (default 2026) reseeds the random parts of the content."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces
from oracle import pyoracle          # checker

mp = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
CHECKER = sys.argv[2] if len(sys.argv) > 2 else "oracle"
if CHECKER == "ref":
    from oracle import pyref
W = 2048
H = max(4, int(mp * 1e6 / W) // 4 * 4)
CSEED = int(os.environ.get("CAMPAIGN_SEED", "2026"))           # another seed = other smooth fields, random bytes, alpha mixes
rng = np.random.default_rng(CSEED)
OFF = CSEED - 2026

def posterised(h, w, levels):
    img = surfaces.ldr_smooth(h, w, seed=surfaces.SEED + 17 + OFF)
    step = 256 // levels
    return ((img // step) * step + step // 2).astype(np.uint8)

def mixed_ldr(h, w):
    q = h // 5 // 4 * 4
    # natural images (the reference's samples among them): the bounded BC7 order visits nearly every block there, bails out of its bound
    # loop, and ties between modes are the content's own
    z, z2 = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz")), np.load(os.path.join(ROOT, "tests", "golden", "samples2.npz"))
    tiles = [z["baboon"], z["monkey"][:216, :216], z2["normals"], z2["landscape_detail"][:336, :124], z2["gradients"], z2["test_a"][:256, :256]]
    nat = np.zeros((q, w, 4), np.uint8)
    x = 0
    while x < w:
        for t in tiles:
            if x >= w:
                break
            tw = min(t.shape[1], w - x)
            col = np.tile(t[:, :tw], (-(-q // t.shape[0]), 1, 1))[:q]
            nat[:, x:x + tw] = col
            x += tw
    parts = [surfaces.ldr_smooth(q, w, seed=surfaces.SEED + 31 + OFF).copy(), rng.integers(0, 256, size=(q, w, 4), dtype=np.uint8),
             posterised(q, w, 4), nat, posterised(h - 4 * q, w, 2)]
    # alpha of the smooth quarter: real alpha | opaque | 254/255 speckles | per-block mix -- the RGBA profiles' order of
    # mode groups and the skipped RGB scans (bc7_finish_all) see whole waves of each kind and mixed ones
    a = parts[0][..., 3]
    a[:, w // 4:w // 2] = 255
    a[:, w // 2:3 * w // 4] = np.where(rng.random((q, w // 4)) < 0.06, 254, 255).astype(np.uint8)
    kinds = np.repeat(np.repeat(rng.integers(0, 3, (q // 4, w // 16)), 4, 0), 4, 1)
    a[:, 3 * w // 4:] = np.where(kinds == 0, 255, np.where(kinds == 1, 254, a[:, 3 * w // 4:])).astype(np.uint8)
    return np.ascontiguousarray(np.concatenate(parts, axis=0))

def mixed_hdr(h, w):
    q = h // 2 // 4 * 4
    a = surfaces.hdr_smooth(q, w, seed=surfaces.SEED + 41 + OFF)
    b = rng.integers(0, 65536, size=(h - q, w, 4), dtype=np.uint16)     # NaN / inf / negative halves included
    return np.ascontiguousarray(np.concatenate([a, b], axis=0))

cases = [("bc1", None), ("bc3", None), ("bc4", None), ("bc5", None)] + [("bc7", p) for p in itw_amd.BC7_PROFILES] + [("bc6h", p) for p in itw_amd.BC6H_PROFILES]
if len(sys.argv) > 3:
    cases = [c for c in cases if c[0] in sys.argv[3].split(",")]
if len(sys.argv) > 4:                                        # profile filter, e.g. slow,alpha_slow
    cases = [c for c in cases if c[1] in sys.argv[4].split(",")]
torch.cuda.set_device(0)
bad_total, blocks_total = 0, 0
ldr, hdr = mixed_ldr(H, W), mixed_hdr(H, W)
for fmt, prof in cases:
    img = hdr if fmt == "bc6h" else ldr
    t0 = time.perf_counter()
    got = itw_amd.compress(fmt, torch.from_numpy(img.view(np.int16) if fmt == "bc6h" else img).cuda(), prof)
    torch.cuda.synchronize()
    got = got.cpu().numpy()
    t1 = time.perf_counter()
    use_ref = CHECKER == "ref" and fmt not in ("bc4", "bc5")
    want = (pyref.encode_mt(fmt, img, prof) if use_ref else pyoracle.encode_mt(fmt, img, prof)).reshape(-1)
    t2 = time.perf_counter()
    bpb = itw_amd.BYTES_PER_BLOCK[fmt]
    bad = int((got.reshape(-1, bpb) != want.reshape(-1, bpb)).any(axis=1).sum())
    n = got.size // bpb
    bad_total += bad; blocks_total += n
    print(f"{fmt:5s} {prof or '-':16s} {n:8d} blocks  mismatches {bad:6d}   gpu {1e3*(t1-t0):8.1f} ms  {'kernel.ispc' if use_ref else 'oracle'} {t2-t1:6.1f} s", flush=True)
print(f"TOTAL {blocks_total} blocks, {bad_total} mismatches (checker: {'the reference kernel.ispc, scalar build' if CHECKER == 'ref' else 'oracle'})")
sys.exit(1 if bad_total else 0)
