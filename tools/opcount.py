"""Measured operation counts of the reference algorithm (via the CPU oracle), per block and per preset.
MEASUREMENT INFRASTRUCTURE: uses oracle/ (never the product).  See tools/opcount/opcount.c.

For each workload the oracle encodes a crop of the bench surface (I3 / I4s) under ptrace single-stepping; executed
instruction addresses inside liboracle_bcn.so are joined with `objdump -d` and classified.  Output: a table on stdout
and profiles/op_counts.json (read by bench.py for roofline.valu.algorithmic_*).

    python tools/opcount.py [--blocks 64] [--workloads bc1,bc7_slow,...]
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np                      # noqa: E402
from itw_amd import surfaces           # noqa: E402

LIB = os.path.join(ROOT, "oracle", "liboracle_bcn.so")
EXE = os.path.join(ROOT, "tools", "opcount", "opcount")
WORKLOADS = {"bc1": ("bc1", "-"), "bc3": ("bc3", "-"),
             "bc7_ultrafast": ("bc7", "ultrafast"), "bc7_veryfast": ("bc7", "veryfast"), "bc7_basic": ("bc7", "basic"),
             "bc7_slow": ("bc7", "slow"), "bc7_alpha_basic": ("bc7", "alpha_basic"), "bc7_alpha_slow": ("bc7", "alpha_slow"),
             "bc6h_fast": ("bc6h", "fast"), "bc6h_slow": ("bc6h", "slow")}

CLASSES = [
    ("fp_mul", r"^v?mul[sp]s$"), ("fp_add", r"^v?(add|sub)[sp]s$"), ("fp_div", r"^v?div[sp]s$"), ("fp_sqrt", r"^v?sqrt[sp]s$"),
    ("fp_cmp_minmax", r"^v?(u?comiss|minss|maxss|minps|maxps|cmp\w*ss|cmp\w*ps)$"),
    ("cvt", r"^v?cvt"), ("fp_move_logic", r"^v?(mov[sdaulhq]*p?s|movd|movq|andn?ps|orps|xorps|pxor|unpck\w+|shufps|pshufd|punpck\w+|movaps|movups|movap[sd])$"),
]


def disassemble():
    out = subprocess.run(["objdump", "-d", "--no-show-raw-insn", LIB], capture_output=True, text=True, check=True).stdout
    ins = {}
    func = None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            func = m.group(1)
            continue
        m = re.match(r"^\s*([0-9a-f]+):\s+(\S+)", line)
        if m:
            ins[int(m.group(1), 16)] = (m.group(2), func)
    return ins


def classify(mn):
    for name, pat in CLASSES:
        if re.match(pat, mn):
            return name
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16, help="blocks per crop for the light formats (bc1, bc3); three crops per workload")
    ap.add_argument("--heavy-blocks", type=int, default=4, help="blocks per crop for bc7 / bc6h (single-stepping runs at ~12 k instructions/s here)")
    ap.add_argument("--workloads", default="bc1,bc3,bc7_veryfast,bc7_basic,bc7_slow,bc7_alpha_slow,bc6h_fast,bc6h_slow")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "op_counts.json"))
    a = ap.parse_args()
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "-s"], check=True)
    subprocess.run(["gcc", "-O2", "-o", EXE, os.path.join(ROOT, "tools", "opcount", "opcount.c"), "-ldl"], check=True)
    ins = disassemble()
    ldr = surfaces.ldr_smooth(4096, 4096)
    hdr = surfaces.hdr_smooth(1024, 1024)
    tmp = tempfile.mkdtemp()
    result = {"method": "ptrace single-step of the scalar C oracle (gcc -O2 -ffp-contract=off, SSE scalar) + objdump classification; "
                        "fp32 ops = mulss/addss/subss/divss/sqrtss, compares = comiss/minss/maxss, conversions = cvt*",
              "sample": f"three square crops of the bench surfaces per workload: {a.blocks} blocks each for bc1/bc3, {a.heavy_blocks} for bc7/bc6h", "workloads": {}}
    print(f"{'workload':<16} {'blocks':>6} {'x86 instr/blk':>14} {'fp mul':>9} {'fp add/sub':>10} {'div':>6} {'sqrt':>6} {'cmp/min/max':>11} {'cvt':>8} {'fp32 ops/blk':>13}")
    for wl in a.workloads.split(","):
        fmt, prof = WORKLOADS[wl]
        side = int(round((a.blocks if fmt in ("bc1", "bc3") else a.heavy_blocks) ** 0.5)) * 4
        tot = collections.Counter()
        nblocks = 0
        t0 = time.time()
        for (y, x) in ((1024, 512), (2048, 3000), (3500, 1800)):          # three places of the surface: different content
            src = hdr if fmt == "bc6h" else ldr
            y, x = y % (src.shape[0] - side), x % (src.shape[1] - side)
            crop = np.ascontiguousarray(src[y:y + side, x:x + side])
            raw, hist = os.path.join(tmp, "in.raw"), os.path.join(tmp, "hist.txt")
            crop.tofile(raw)
            subprocess.run([EXE, LIB, fmt, prof, str(side), str(side), raw, hist], check=True, timeout=7200)
            for line in open(hist):
                if line.startswith("#"):
                    continue
                off, cnt = line.split()
                mn = ins.get(int(off, 16), ("?", None))[0]
                tot[classify(mn)] += int(cnt)
                tot["x86_instructions"] += int(cnt)
            nblocks += (side // 4) ** 2
        per = {k: v / nblocks for k, v in tot.items()}
        fp = per.get("fp_mul", 0) + per.get("fp_add", 0) + per.get("fp_div", 0) + per.get("fp_sqrt", 0)
        per["fp32_arith_ops"] = fp
        per["fp32_arith_cmp_cvt_ops"] = fp + per.get("fp_cmp_minmax", 0) + per.get("cvt", 0)
        result["workloads"][wl] = {"blocks": nblocks, "per_block": {k: round(v, 1) for k, v in per.items()}, "seconds": round(time.time() - t0, 1)}
        print(f"{wl:<16} {nblocks:>6} {per['x86_instructions']:>14.0f} {per.get('fp_mul', 0):>9.0f} {per.get('fp_add', 0):>10.0f} {per.get('fp_div', 0):>6.1f} "
              f"{per.get('fp_sqrt', 0):>6.1f} {per.get('fp_cmp_minmax', 0):>11.0f} {per.get('cvt', 0):>8.0f} {fp:>13.0f}", flush=True)
        with open(a.out, "w") as f:
            json.dump(result, f, indent=1)


if __name__ == "__main__":
    main()
