"""CPU feasibility study: exact pruning of BC6H `slow`'s two-region scans.

Under the slow profiles each of the six two-region modes (0, 1, 2, 5, 6, 9) scans all 32 shapes in ascending order of the ranking key
(int)(PCA residual bound) (kernel.ispc:2257-2273) and keeps the first strict minimum of bc6h_enc_2p_part_fast's error (:2195-2216).  A shape
whose exact lower bound -- sum over its two subsets of (sqrt(R_s) - sqrt(3)/2 sqrt(n_s))_+^2, R_s = the subset's residual about its best
line in the uf16 domain the encoder works in (float64, exact eigenvalue; every decoded level is a rounded point of one segment per subset,
:1164-1170) -- is >= the mode's best error so far cannot replace it and could be skipped.  This script takes the reference-order scan from
the oracle (oracle_bc6h_2p_scan: ranked list, key, part_fast error per position, per mode) and counts what such a rule would skip:
  * per (block, mode, position): skippable visits;
  * per (wave of 64 consecutive blocks, mode, position): visits NO lane needs (what a SIMT kernel saves without compaction);
  * per (block, position): shapes no mode needs (the shared line fit of the product's kernel, one per shape for all six modes).
Blocks that can reach the S5 int-overflow quirk (kernel.ispc:1178; a channel span above 26 754 in the uf16 domain, where "errors" wrap
negative) are exempt: all their visits count as needed.
"""
import ctypes as C, os, sys, re
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
from oracle import pyoracle
from itw_amd import surfaces

MODES = (0, 1, 2, 5, 6, 9)


def masks32():
    t = open(os.path.join(ROOT, "oracle", "bc7_tables.h")).read()
    m = re.search(r"BCN_SUBSET_MASKS\[128\]\s*=\s*\{([^}]*)\}", t)
    return [int(x, 16) & 0xffff for x in re.findall(r"0x([0-9a-fA-F]+)u", m.group(1))][:32]


def study(name, img, out, limit_blocks=8192):
    L = pyoracle.lib()
    L.oracle_bc6h_2p_scan.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_bc6h_2p_scan.restype = C.c_float
    L.oracle_bc6h_setup_block.argtypes = [C.c_void_p, C.c_void_p]
    h, w = img.shape[0] // 4 * 4, img.shape[1] // 4 * 4
    t = img[:h, :w].astype(np.float32).reshape(h // 4, 4, w // 4, 4, 4).transpose(0, 2, 4, 1, 3).reshape(-1, 64)
    t = np.ascontiguousarray(t[:limit_blocks])
    nb = t.shape[0] // 64 * 64
    t = t[:nb]
    masks = masks32()
    sel = [np.array([(m >> k) & 1 for k in range(16)], bool) for m in masks]
    # exact bounds per block and shape, in the encoder's domain
    dom = np.zeros_like(t)
    for b in range(nb):
        L.oracle_bc6h_setup_block(t[b].ctypes.data, dom[b].ctypes.data)
    tex = dom[:, :48].reshape(nb, 3, 16).transpose(0, 2, 1).astype(np.float64)
    lb = np.zeros((nb, 32))
    for p in range(32):
        for s in (sel[p], ~sel[p]):
            x = tex[:, s, :]
            x = x - x.mean(axis=1, keepdims=True)
            c = np.einsum("bki,bkj->bij", x, x)
            r = np.maximum(np.trace(c, axis1=1, axis2=2) - np.linalg.eigvalsh(c)[:, -1], 0)
            lb[:, p] += np.maximum(np.sqrt(r) - np.sqrt(3) / 2 * np.sqrt(s.sum()), 0) ** 2
    span = (tex.max(axis=1) - tex.min(axis=1)).max(axis=1)
    exempt = span > 26754
    need = np.ones((nb, len(MODES), 32), bool)          # [block][mode][position]
    pos_shape = np.zeros((nb, len(MODES), 32), np.int32)
    lst = (C.c_int32 * 32)(); key = (C.c_int32 * 32)(); err = (C.c_float * 32)()
    violations = 0
    for b in range(nb):
        for mi, mode in enumerate(MODES):
            L.oracle_bc6h_2p_scan(t[b].ctypes.data, mode, lst, key, err)
            best = np.inf
            for i in range(32):
                p = lst[i]
                pos_shape[b, mi, i] = p
                if not exempt[b]:
                    if lb[b, p] >= best:
                        need[b, mi, i] = False
                    if err[i] < lb[b, p] * (1 - 1e-6) - 1e-3:
                        violations += 1
                if err[i] < best:
                    best = err[i]
    per_visit = 1 - need.mean()
    nw = nb // 64
    wave_need = need.reshape(nw, 64, len(MODES), 32).any(axis=1)
    per_wave = 1 - wave_need.mean()
    # shape-level: the product fits each shape once for all six modes, in ranked order (the ranking is mode independent)
    shape_need = need.any(axis=1)                         # [block][position]
    per_shape = 1 - shape_need.mean()
    wave_shape = 1 - shape_need.reshape(nw, 64, 32).any(axis=1).mean()
    first_stop = np.array([[np.max(np.nonzero(need[b, mi])[0]) + 1 for mi in range(len(MODES))] for b in range(nb)])
    line = (f"{name:16s} blocks {nb:6d} exempt (S5) {100 * exempt.mean():5.1f} % | skippable (block, mode, position) visits {100 * per_visit:5.1f} %  "
            f"per wave of 64: {100 * per_wave:5.1f} % | shapes no mode needs: per block {100 * per_shape:5.1f} %, per wave {100 * wave_shape:5.1f} % | "
            f"mean last needed position {first_stop.mean():5.1f} of 32 | bound above an actual error: {violations}")
    print(line, flush=True)
    out.append(line)


if __name__ == "__main__":
    out = []
    g = os.path.join(ROOT, "tests", "golden")
    z = np.load(os.path.join(g, "inputs.npz")); z2 = np.load(os.path.join(g, "samples2.npz"))
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    study("I4 monkey_hdr", z["monkey_hdr"], out, n)
    study("I4s hdr_smooth", surfaces.hdr_smooth(256, 256), out, n)
    study("hdr_probe", z2["hdr_probe"], out, n)
    study("I4r random bits", surfaces.hdr_random_bits(64, 256), out, n)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            f.write(__doc__ + "\n" + "\n".join(out) + "\n")
