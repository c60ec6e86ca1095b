"""Issue-cycle model of a kernel: its ISA, basic block by basic block, weighted by how often each block runs per wave
("static opcode histogram weighted by loop trip counts; bench.py emits roofline.issue").

    python tools/isa_weighted.py <spec.json> [--dump]        -> profiles/<tag>_isa_<kernel>.txt + profiles/<tag>_issue_model.json

How a block's weight is found (executions per wave):
  * the compiler's own loop annotations in the assembly ("Loop Header: Depth=n", "in Loop: Header=BBx_y") give every block its
    enclosing loops; the spec names each loop header's trip count (the loops of the `slow` scans are wave-uniform: 64 shapes,
    2 or 3 subsets, schedule-driven cache hits);
  * a block entered through `s_bitcmp{0,1}_b32 <mask>, k` + `s_cbranch_scc` is the body of texel k of a subset: it runs for the
    fraction of texels a subset holds (1 / subsets of the shape, exact on average: the subsets of a shape partition 16 texels);
  * blocks behind a lane-dependent branch (s_cbranch_vccz/vccnz/execz/execnz) are the rare paths (NaN probes, exact ties):
    weight `rare` of the spec (default 0);
  * anything else the spec overrides by label: wave-uniform branches on kernel arguments (a mode switched off = 0) and on the
    scan schedule (cache hit fractions).
Issue cost per opcode: profiles/valu_issue_costs.md (tools/ubench): 2 cycles for the plain VOP2 forms, 8 for v_rcp / v_sqrt /
64-bit multiplies, 4 for everything else.  The model's VALU instruction total per wave is printed next to the SQ_INSTS_VALU
measurement named in the spec: that ratio is the check on the weights.
"""
import collections
import hashlib
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "intel-texture-works-plugin_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-flush-denormals-to-zero",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "--cuda-device-only", "-S"]
PER_SOURCE = {"bc7.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}      # csrc/Makefile EXTRA_<stem>
TWO = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_mov_b64",
       "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_cndmask_b32", "v_add_u16", "v_sub_u16",
       "v_mul_lo_u16", "v_ashrrev_i16", "v_mul_f16", "v_mul_legacy_f32"}
EIGHT = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32", "v_mad_u64_u32", "v_mul_hi_u32"}


def cost(op):
    return 8 if op in EIGHT else 2 if op in TWO else 4


def source_sha256():
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".hpp", ".h")) or name == "Makefile":
            h.update(name.encode() + b"\0")
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def assembly(src, cache_dir="/tmp/isa_weighted"):
    os.makedirs(cache_dir, exist_ok=True)
    key = hashlib.sha256((source_sha256() + src).encode()).hexdigest()[:16]
    out = os.path.join(cache_dir, f"{os.path.basename(src)}.{key}.s")
    if not os.path.exists(out):
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + PER_SOURCE.get(src, []) + ["-o", out, os.path.join(CSRC, src)], check=True,
                       stderr=subprocess.DEVNULL, cwd=CSRC)
    return open(out).read().split("\n")


def kernel_body(text, pattern):
    """Lines of the first kernel whose demangled name contains `pattern`."""
    cur, name = None, None
    for line in text:
        m = re.match(r"^(_Z\w+):", line)
        if m and cur is None:
            dem = subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE, text=True).stdout.strip()
            if pattern in dem:
                name, cur = dem.replace("(anonymous namespace)::", "").split("(")[0], []
                continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):             # (an early `return` puts an s_endpgm in the middle of the function)
                return name, cur
            cur.append(line)
    raise SystemExit(f"no kernel matching {pattern!r}")


class Block:
    def __init__(self, label):
        self.label, self.loops, self.ops, self.lines = label, [], collections.Counter(), []
        self.header_of = None          # label of the loop this block heads
        self.parents, self.inner = [], None
        self.guard = None              # "texel" | "rare" | None: how the block is entered
        self.last_branch = None


def parse_blocks(body):
    blocks, cur = [], Block("entry")
    blocks.append(cur)
    pending_comments = []
    for line in body:
        m = re.match(r"^\.(LBB\d+_\d+):(.*)$", line) or re.match(r"^; %bb\.(\d+):(.*)$", line)
        if m:
            label = m.group(1) if m.group(1).startswith("LBB") else "bb." + m.group(1)
            cur = Block(label)
            blocks.append(cur)
            pending_comments = [m.group(2)]
            cur.lines.append(line)
            continue
        cur.lines.append(line)
        if re.match(r"^\s*;", line) and not cur.ops and not re.search(r"^\s+[sv]_", line):
            pending_comments.append(line)
            continue
        m = re.match(r"\s+([a-z][a-z0-9_]+)", line)
        if m:
            op = re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", m.group(1))
            if op.startswith("v_"):
                cur.ops[op] += 1
            if op.startswith("s_cbranch") or op == "s_branch":
                cur.last_branch = line.strip()
        for c in pending_comments:
            if re.search(r"Loop Header: Depth=(\d+)", c):
                cur.header_of = cur.label
            for ml in re.finditer(r"Parent Loop\s*(BB\d+_\d+)", c):
                cur.parents.append("L" + ml.group(1))
            ml = re.search(r"in Loop: Header=(BB\d+_\d+)", c)
            if ml:
                cur.inner = "L" + ml.group(1)               # the innermost loop the block belongs to (its header's label)
        pending_comments = []
    # a block's loops, outermost first: its innermost loop's parents (named on that loop's header), then the loop itself
    headers = {b.label: b for b in blocks if b.header_of}
    for b in blocks:
        inner = b.label if b.header_of else b.inner
        if inner and inner in headers:
            b.loops = list(headers[inner].parents) + [inner]
    return blocks


def mark_guards(blocks):
    """Texel bodies (entered through s_bitcmp + s_cbranch_scc on a mask bit) and rare paths (lane-dependent branches)."""
    by_label = {b.label: b for b in blocks}
    for i, b in enumerate(blocks):
        text = [l.strip() for l in b.lines if re.match(r"\s+[sv]_", l)]
        for j, l in enumerate(text):
            m = re.match(r"s_cbranch_(scc0|scc1|vccz|vccnz|execz|execnz)\s+\.(LBB\d+_\d+)", l)
            if not m:
                continue
            kind, target = m.group(1), m.group(2)
            nxt = blocks[i + 1] if i + 1 < len(blocks) else None
            prev = text[j - 1] if j else ""
            mb = re.match(r"s_bitcmp([01])_b32\s+s\d+,\s*(\d+)", prev)
            if kind in ("scc0", "scc1") and mb:
                bit_is_one_on_taken = (mb.group(1) == "0") == (kind == "scc0")     # bitcmp0 sets SCC when the bit is 0
                body = by_label.get(target) if bit_is_one_on_taken else nxt
                if body is not None and body.guard is None:
                    body.guard = "texel"
            elif kind in ("vccz", "vccnz", "execz", "execnz"):
                # the out-of-line side of a lane-dependent branch: the target when it is not simply the next block
                t = by_label.get(target)
                if t is not None and t is not nxt and t.guard is None and j == len(text) - 1:
                    pass                                   # a skip-ahead: what is skipped (the fallthrough) is the conditional part
                if nxt is not None and nxt.guard is None and kind in ("execz", "vccz", "vccnz", "execnz") and j == len(text) - 1:
                    nxt.guard = "rare?"                    # resolved by the spec's `rare` list / default below


def segments(blocks):
    """Cuts the kernel into SEGMENTS at its groups of texel bodies: a segment = the blocks after the previous texel group up to and
    including the next one (a subset's fit ends with its extreme projections over the texels; each mode's part ends with its texel
    errors).  A texel group = consecutive texel bodies of equal size; the first / last texel of a group is often laid out without
    the bit test (the compiler folds it into a vector condition), so equally sized neighbours join the group.  The spec weights
    whole segments (a mode that is off, a cache hit rate), which keeps it independent of the compiler's block numbering."""
    seg_of, groups, cur, i, n = {}, [], 0, 0, len(blocks)
    sizes = [sum(b.ops.values()) for b in blocks]
    while i < n:
        b = blocks[i]
        if b.guard == "texel":
            size, j, members = sizes[i], i, []
            while j < n and (blocks[j].guard == "texel" and sizes[j] == size or sizes[j] <= 2 and blocks[j].guard != "texel" and j + 1 < n and blocks[j + 1].guard == "texel" and sizes[j + 1] == size):
                if blocks[j].guard == "texel":
                    members.append(j)
                j += 1
            # neighbours of the same size without the bit test: first / last texel of the group
            k = members[0] - 1
            while k >= 0 and sizes[k] <= 2:
                k -= 1
            if k >= 0 and blocks[k].guard != "texel" and abs(sizes[k] - size) <= 2 and sizes[k] > 2:
                members.insert(0, k)
            k = j
            while k < n and sizes[k] <= 2:
                k += 1
            if k < n and blocks[k].guard != "texel" and abs(sizes[k] - size) <= 2:
                members.append(k)
                j = k + 1
            for m in members:
                blocks[m].guard = "texel"
            lo = min(members[0], i)
            for m in range(lo, j):
                seg_of[m] = cur
            groups.append((cur, size, len(members), blocks[members[0]].label))
            cur += 1
            i = j
        else:
            seg_of[i] = cur
            i += 1
    return seg_of, groups


def main():
    spec_path = sys.argv[1]
    dump = "--dump" in sys.argv
    spec = json.load(open(spec_path))
    tag = spec.get("tag", "r04")
    text_cache = {}
    model = {"_source_sha256": source_sha256(), "_how": "tools/isa_weighted.py " + os.path.relpath(spec_path, ROOT), "kernels": {}, "workloads": {}}
    for k in spec["kernels"]:
        src = k["source"]
        if src not in text_cache:
            text_cache[src] = assembly(src)
        name, body = kernel_body(text_cache[src], k["match"])
        blocks = parse_blocks(body)
        mark_guards(blocks)
        seg_of, groups = segments(blocks)
        seg_w = k.get("segments", {})                         # segment ordinal (as a string) -> factor; default 1
        sig_rare = k.get("rare_signatures", [])               # a block holding one of these instruction texts is a rare path: weight 0
        trips = dict(k.get("loops", {}))
        order = [b.label for b in blocks if b.header_of]      # loop headers in assembly order
        for i, t in enumerate(k.get("loops_by_order", [])):
            if i < len(order):
                trips.setdefault(order[i], t)
        seg_block = k.get("seg_block", {})                    # "<segment>:<n>" -> weight of the segment's n-th non-texel block inside the loops
        seen_in_seg = collections.Counter()
        over = k.get("blocks", {})
        texel_frac = k.get("texel_fraction", {})              # loop label -> fraction; "default" for the rest
        rare_w = k.get("rare", 0.0)
        rows, tot_ops, tot_cyc = [], collections.Counter(), 0.0
        unknown_loops = set()
        for bi, b in enumerate(blocks):
            sg = seg_of.get(bi, -1)
            w = seg_w.get(str(sg), 1.0)
            if b.guard != "texel" and len(b.loops) >= 2 and sum(b.ops.values()):
                key = f"{sg}:{seen_in_seg[sg]}"
                seen_in_seg[sg] += 1
                w = seg_block.get(key, w)
            if sig_rare and any(sig in l for sig in sig_rare for l in b.lines):
                w = 0.0
            for key in b.loops:
                if key in trips:
                    w *= trips[key]
                else:
                    unknown_loops.add(key)
            if b.label in over:
                w *= over[b.label]
            elif b.guard == "texel":
                inner = b.loops[-1] if b.loops else None
                w *= texel_frac.get(inner, texel_frac.get("default", 0.5))
            elif b.guard == "rare?":
                w *= over.get("rare:" + b.label, 1.0)
            n = sum(b.ops.values())
            cyc = sum(cost(o) * c for o, c in b.ops.items())
            rows.append((b.label, len(b.loops), b.guard or "", seg_of.get(bi, -1), w, n, cyc))
            for o, c in b.ops.items():
                tot_ops[o] += c * w
            tot_cyc += cyc * w
        n_valu = sum(tot_ops.values())
        by_cost = {c: sum(v for o, v in tot_ops.items() if cost(o) == c) for c in (2, 4, 8)}
        out = [f"## {name}   (tools/isa_weighted.py, spec {os.path.relpath(spec_path, ROOT)}, source sha256 {model['_source_sha256'][:16]})",
               f"executed per wave (model): {n_valu:,.0f} VALU instructions = {by_cost[2]:,.0f} two-cycle + {by_cost[4]:,.0f} four-cycle + {by_cost[8]:,.0f} eight-cycle forms"
               f"  -> {tot_cyc:,.0f} issue cycles per wave"]
        meas = k.get("measured_valu_per_wave")
        if meas:
            out.append(f"measured SQ_INSTS_VALU per wave: {meas:,.0f} ({k.get('measured_source', '')})  model / measured = {n_valu / meas:.3f}")
        out.append("loop headers in assembly order and their trip counts: " + "  ".join(f"{h}={trips.get(h, 1)}" for h in order))
        for c in (2, 4, 8):
            row = sorted(((v, o) for o, v in tot_ops.items() if cost(o) == c), reverse=True)
            out.append(f"  {c}-cycle: " + "  ".join(f"{o} {v:,.0f}" for v, o in row if v >= 0.5))
        if dump:
            out.append("  texel groups (segment, instructions per texel body, bodies, first label): " + "  ".join(str(g) for g in groups))
            out.append("  blocks: label depth guard segment weight valu cycles")
            for r in rows:
                if r[5]:
                    out.append("    %-12s d%d %-6s seg%-3d w=%9.3f valu=%4d cyc=%5d" % r)
        path = os.path.join(ROOT, "profiles", f"{tag}_isa_{k['name']}.txt")
        with open(path, "w") as f:
            f.write("\n".join(out) + "\n")
        print("\n".join(out[:4]))
        model["kernels"][k["name"]] = {"kernel": name, "valu_per_wave": round(n_valu), "issue_cycles_per_wave": round(tot_cyc),
                                       "two_cycle": round(by_cost[2]), "four_cycle": round(by_cost[4]), "eight_cycle": round(by_cost[8]),
                                       "cycles_per_instruction": round(tot_cyc / max(n_valu, 1), 4),
                                       "weights": "loop trip counts + segment weights" if k.get("loops_by_order") or k.get("loops") else "static (every block once)",
                                       "measured_valu_per_wave": meas, "file": f"profiles/{tag}_isa_{k['name']}.txt"}
        # what bench.py looks up: by the kernel's name as rocprofv3 prints it (a kernel that runs as several kinds of waves -- the two
        # families of bc7_scan_all -- is the sum of its parts)
        agg = model.setdefault("by_rocprof_name", {}).setdefault(name.replace("void ", ""), {"valu": 0.0, "cycles": 0.0, "parts": []})
        agg["valu"] += n_valu; agg["cycles"] += tot_cyc; agg["parts"].append(k["name"])
        agg["cycles_per_instruction"] = round(agg["cycles"] / max(agg["valu"], 1), 4)
    summary = ["# executed VALU instructions per unit of work: weighted ISA model vs the SQ_INSTS_VALU counter (tools/isa_weighted.py)",
               "# kernel | model instructions | measured | model / measured | cycles per instruction (what bench.py's roofline.issue uses)"]
    for kname, meas in spec.get("measured", {}).items():
        agg = model["by_rocprof_name"].get(kname)
        if not agg:
            continue
        agg["measured_valu_per_unit"] = meas["valu_per_unit"]
        agg["model_over_measured"] = round(agg["valu"] / meas["valu_per_unit"], 3)
        agg["measured_unit"], agg["measured_source"] = meas["unit"], meas["source"]
        summary.append(f"{kname} | {agg['valu']:,.0f} | {meas['valu_per_unit']:,} per {meas['unit']} | {agg['model_over_measured']} | {agg['cycles_per_instruction']}")
    with open(os.path.join(ROOT, "profiles", f"{tag}_isa_summary.txt"), "w") as f:
        f.write("\n".join(summary) + "\n")
    print("\n".join(summary))
    for wl, parts in spec.get("workloads", {}).items():
        cyc = sum(model["kernels"][p["kernel"]]["issue_cycles_per_wave"] * p["waves_per_call"] for p in parts)
        valu = sum(model["kernels"][p["kernel"]]["valu_per_wave"] * p["waves_per_call"] for p in parts)
        model["workloads"][wl] = {"issue_cycles_per_call": int(cyc), "valu_wave_instructions_per_call": int(valu), "parts": parts,
                                  "surface": "4096x4096 (1 048 576 blocks, 16 384 block-waves)"}
    with open(os.path.join(ROOT, "profiles", f"{tag}_issue_model.json"), "w") as f:
        json.dump(model, f, indent=1)
    print("wrote profiles/%s_issue_model.json" % tag)


if __name__ == "__main__":
    main()
