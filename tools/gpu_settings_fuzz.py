"""GPU fuzz: random bc7_enc_settings / bc6h_enc_settings structs x adversarial block classes, HIP kernels (both BC7 launch
shapes) against the reference's own kernel.ispc (scalar build, oracle/_ref/libispc_texcomp_ref_full.so; the oracle when
that library is absent).  Usage: python tools/gpu_settings_fuzz.py [trials] [seed]; exit 1 on the first mismatch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np, torch
import itw_amd
from itw_amd import surfaces
from oracle import pyoracle, pyref          # checkers

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
check = pyref if pyref.available() else pyoracle
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)


def ldr_classes(n):
    w = 4 * n
    flat = np.repeat(np.repeat(rng.integers(0, 256, (1, n, 4), dtype=np.uint8), 4, 0), 4, 1)
    two = np.where(rng.random((4, w, 1)) < 0.5, np.repeat(rng.integers(0, 256, (1, n, 4)), 4, 1), np.repeat(rng.integers(0, 256, (1, n, 4)), 4, 1)).astype(np.uint8)
    x = np.linspace(0, 255, w)[None, :, None] * np.ones((4, 1, 4))
    grad = (x * rng.random((1, 1, 4)) + np.arange(4)[:, None, None] * 9).clip(0, 255).astype(np.uint8)
    ext = rng.choice(np.array([0, 1, 127, 128, 254, 255], dtype=np.uint8), (4, w, 4))
    noise = (x + rng.integers(-6, 7, (4, w, 4))).clip(0, 255).astype(np.uint8)
    cut = surfaces.ldr_smooth(4, w, seed=int(rng.integers(1, 1 << 20))).copy()
    cut[..., 3] = np.where(rng.random((4, w)) < 0.5, 0, 255)
    near = surfaces.ldr_smooth(4, w, seed=int(rng.integers(1, 1 << 20))).copy()
    near[..., 3] = rng.integers(250, 256, (4, w))
    opaque = surfaces.ldr_smooth(4, w, seed=int(rng.integers(1, 1 << 20))).copy()
    opaque[..., 3] = 255
    return np.ascontiguousarray(np.concatenate([flat, two, grad, ext, noise, cut, near, opaque, rng.integers(0, 256, (4, w, 4), dtype=np.uint8)], axis=0))


def hdr_classes(n):
    w = 4 * n
    smooth = surfaces.hdr_smooth(4, w, seed=int(rng.integers(1, 1 << 20))).view(np.uint16)
    bits = rng.integers(0, 65536, (4, w, 4), dtype=np.uint16)
    flat = np.repeat(np.repeat(rng.integers(0, 0x7c00, (1, n, 4), dtype=np.uint16), 4, 0), 4, 1)
    small = rng.integers(0, 64, (4, w, 4), dtype=np.uint16)
    big = rng.integers(0x7800, 0x7c00, (4, w, 4), dtype=np.uint16)
    narrow = (np.uint16(0x3c00) + rng.integers(0, 40, (4, w, 4))).astype(np.uint16)
    return np.ascontiguousarray(np.concatenate([smooth, bits, flat, small, big, narrow], axis=0))


thr = [0, 1, 2, 3, 5, 12, 16, 17, 33, 63, 64, 70]
blocks = 0
for t in range(trials):
    img = ldr_classes(64)                                   # 64 blocks per class row = one wave per class: whole waves of each kind
    s, so = itw_amd.Bc7Settings(), pyoracle.Bc7Settings()
    vals = {"skip_mode2": bool(rng.integers(0, 2)), "fastSkipTreshold_mode1": int(rng.choice(thr)), "fastSkipTreshold_mode3": int(rng.choice(thr)),
            "fastSkipTreshold_mode7": int(rng.choice(thr)), "mode45_channel0": int(rng.integers(0, 4)),
            "refineIterations_channel": int(rng.integers(0, 6)), "channels": int(rng.choice([3, 4]))}
    sel = [bool(rng.integers(0, 2)) for _ in range(4)]
    if not any(sel): sel[int(rng.integers(0, 4))] = True
    if rng.random() < 0.35:
        # the bounded mode order (csrc/bc7.hip: modes 1/3, and 7 under RGBA profiles, last and only where their bound allows) needs both
        # multi-subset groups on and whole-table scans: a third of the trials are drawn from there
        sel[0] = sel[1] = True
        vals["fastSkipTreshold_mode1"] = int(rng.choice([0, 64, 64, 70])); vals["fastSkipTreshold_mode3"] = int(rng.choice([0, 64, 64, 70]))
        vals["fastSkipTreshold_mode7"] = int(rng.choice([0, 5, 64, 64, 70]))
    ref = [int(rng.integers(0, 6)) for _ in range(8)]
    for x in (s, so):
        for k, v in vals.items(): setattr(x, k, v)
        for i in range(4): x.mode_selection[i] = sel[i]
        for i in range(8): x.refineIterations[i] = ref[i]
    want = check.encode("bc7", img, so).reshape(-1)
    d_img = torch.from_numpy(img).to(dev)
    for path in ("deep", "wide"):
        itw_amd.set_bc7_path(path)
        got = itw_amd.compress("bc7", d_img, s); torch.cuda.synchronize()
        if not np.array_equal(got.cpu().numpy().reshape(-1), want):
            bad = np.nonzero((got.cpu().numpy().reshape(-1, 16) != want.reshape(-1, 16)).any(axis=1))[0]
            print("BC7 MISMATCH trial", t, path, "blocks", bad[:8], vals, sel, ref); sys.exit(1)
    itw_amd.set_bc7_path("auto")
    h = hdr_classes(64)
    s6, s6o = itw_amd.Bc6hSettings(), pyoracle.Bc6hSettings()
    v6 = {"slow_mode": bool(rng.integers(0, 2)), "fast_mode": bool(rng.integers(0, 2)), "refineIterations_1p": int(rng.integers(0, 4)),
          "refineIterations_2p": int(rng.integers(0, 4)), "fastSkipTreshold": int(rng.choice([0, 1, 2, 4, 10, 31, 32]))}
    for x in (s6, s6o):
        for k, v in v6.items(): setattr(x, k, v)
    want = check.encode("bc6h", h, s6o).reshape(-1)
    got = itw_amd.compress("bc6h", torch.from_numpy(h.view(np.int16)).to(dev), s6); torch.cuda.synchronize()
    if not np.array_equal(got.cpu().numpy().reshape(-1), want):
        print("BC6H MISMATCH trial", t, v6); sys.exit(1)
    blocks += 2 * (img.shape[0] // 4) * 64 + (h.shape[0] // 4) * 64
print(f"{trials} random bc7_enc_settings (both launch shapes) + {trials} random bc6h_enc_settings structs x 9 LDR / 6 HDR block classes, one wave "
      f"per class: {blocks} blocks, 0 mismatches (checker: {'the reference kernel.ispc, scalar build' if check is pyref else 'oracle'})")
