"""Copies the rocprofv3 summaries of tools/profile_gpu.sh <tag> from gpurun_out/prof_<tag>/ into profiles/<tag>_* and
rebuilds profiles/pmc_traffic.json (HBM bytes per C-ABI call from the FETCH_SIZE / WRITE_SIZE passes, with the gfx950
corrections of MI355X_MICROARCH.md: FETCH_SIZE counts 32 B units reported in KiB at half scale -> x2; WRITE_SIZE as is)
and profiles/<tag>_valu.json (VALU wave-instructions per call from the SQ pass).

Usage: python tools/summarize_profiles.py <tag>
"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


ALL_WORKLOADS = ("bc7_slow", "bc7_alpha_slow", "bc7_basic", "bc7_veryfast", "bc7_alpha_basic", "bc7_alpha_veryfast", "bc6h_slow", "bc6h_fast",
                 "bc1", "bc3", "bc4", "bc5")
SQ_COUNTERS = ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVES")


def short_name(kernel_name):
    return kernel_name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


def abi_calls(src, stem, overrides=None):
    """C-ABI calls the profiled bench command made: `abi_calls` of the JSON line bench.py printed under the profiler
    (gpurun_out/prof_<tag>/<stem>.json, kept by tools/profile_gpu.sh), or `--calls <stem>=N`.  Never inferred from the dispatch counts:
    a call is several launches of the same kernel (two bands on two streams since round 5) and no kernel needs to run exactly once."""
    if overrides and stem in overrides:
        return int(overrides[stem])
    path = os.path.join(src, stem + ".json")
    try:
        with open(path) as f:
            lines = [ln for ln in f.read().splitlines() if ln.startswith("{")]
        n = int(json.loads(lines[-1])["abi_calls"])
    except (OSError, ValueError, KeyError, IndexError):
        raise SystemExit(f"{path}: no bench line with `abi_calls` beside the counter file -- pass --calls {stem}=N")
    if n < 1:
        raise SystemExit(f"{path}: abi_calls = {n}")
    return n


def per_call(path, counter, calls):
    """Sum of `counter` over all kernels of one C-ABI call = total over the trace / `calls` (abi_calls())."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    if not rows:
        return None
    return sum(float(r["Counter_Value"]) for r in rows) / calls


def per_call_by_kernel(path, counter, calls):
    """{kernel name: counter per C-ABI call}."""
    out = collections.defaultdict(float)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[short_name(r["Kernel_Name"])] += float(r["Counter_Value"])
    return {k: v / calls for k, v in out.items()}


def isolated_kernels(path, calls):
    """Per kernel and C-ABI call, from a counter pass (rocprofv3 serialises the dispatches of a --pmc run, so these durations do NOT overlap,
    unlike the kernel-trace summary of a call whose bands share the chip on two streams): {kernel: {ms, dispatches, wave_valu, waves,
    vgprs, lds_bytes}} -- ms = sum of (End - Start) over the kernel's dispatches / calls."""
    seen = {}
    for r in csv.DictReader(open(path)):
        d = seen.setdefault(r["Dispatch_Id"], {"k": short_name(r["Kernel_Name"]), "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                                               "vgpr": int(r["VGPR_Count"]), "lds": int(r["LDS_Block_Size"]), "c": {}})
        d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
    out = {}
    for d in seen.values():
        o = out.setdefault(d["k"], {"ms": 0.0, "dispatches": 0, "wave_valu": 0.0, "waves": 0.0, "vgprs": d["vgpr"], "lds_bytes": d["lds"]})
        o["ms"] += d["ns"] * 1e-6
        o["dispatches"] += 1
        o["wave_valu"] += d["c"].get("SQ_INSTS_VALU", 0.0)
        o["waves"] += d["c"].get("SQ_WAVES", 0.0)
    for o in out.values():
        o["ms"] = round(o["ms"] / calls, 5)
        o["dispatches"] = round(o["dispatches"] / calls, 3)
        o["wave_valu"] = o["wave_valu"] / calls
        o["waves"] = o["waves"] / calls
    return out


def main():
    tag = sys.argv[1]
    overrides = {}
    args = sys.argv[2:]
    while args:
        if args[0] == "--calls" and len(args) > 1:
            k, v = args[1].split("=")
            overrides[k] = int(v)
            args = args[2:]
        elif args[0] == "--src" and len(args) > 1:
            args = args[2:]
        else:
            raise SystemExit("usage: summarize_profiles.py <tag> [--calls <counter file stem>=N ...]")
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    summarize(tag, src, dst, overrides)


def summarize(tag, src, dst, overrides=None):
    os.makedirs(dst, exist_ok=True)
    try:
        stamp = open(os.path.join(src, "source_sha256.txt")).read().strip()
    except OSError:
        stamp = None
    # the kernel-trace summaries stay at the top level (bench.py reads the latest *_kernel_stats.csv); the per-pass counter files and the
    # bench lines they were taken under go to profiles/<tag>_counters/
    raw = os.path.join(dst, f"{tag}_counters")
    for name in sorted(os.listdir(src)):
        if name.startswith("pmc_") and name.endswith((".csv", ".json")):
            os.makedirs(raw, exist_ok=True)
            shutil.copy(os.path.join(src, name), os.path.join(raw, name))
        elif name.endswith((".csv", ".json")):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
            if stamp and name.startswith("kernel_stats"):            # csv files cannot carry the stamp themselves: a sidecar does
                with open(os.path.join(dst, f"{tag}_{name}.sha256"), "w") as fh:
                    fh.write(stamp + "\n")
    if stamp:
        with open(os.path.join(dst, f"{tag}_source_sha256.txt"), "w") as fh:
            fh.write(stamp + "\n")
    traffic = {}
    for wl in ALL_WORKLOADS:
        f = os.path.join(src, f"pmc_{wl}_FETCH_SIZE.csv")
        w = os.path.join(src, f"pmc_{wl}_WRITE_SIZE.csv")
        if not (os.path.exists(f) and os.path.exists(w)):
            continue
        nf, nw = abi_calls(src, f"pmc_{wl}_FETCH_SIZE", overrides), abi_calls(src, f"pmc_{wl}_WRITE_SIZE", overrides)
        fetch_kib = per_call(f, "FETCH_SIZE", nf)
        write_kib = per_call(w, "WRITE_SIZE", nw)
        if fetch_kib is None or write_kib is None:
            continue
        fetch = fetch_kib * 1024 * 2
        write = write_kib * 1024
        traffic[wl] = {"hbm_bytes_per_launch": int(fetch + write), "fetch_bytes_corrected_x2": int(fetch),
                       "write_bytes": int(write), "raw_FETCH_SIZE_KiB": fetch_kib, "raw_WRITE_SIZE_KiB": write_kib,
                       "calls_sampled": nf, "source": f"profiles/{tag}_counters/pmc_{wl}_FETCH_SIZE.csv, pmc_{wl}_WRITE_SIZE.csv"}
    traffic["_source_sha256"] = stamp
    traffic["_note"] = ("rocprofv3 --pmc, one counter per pass (TCC slots); per C-ABI call = all kernels of the call (call count = `abi_calls` of the "
                        "profiled bench line); gfx950 x2 correction applied to FETCH_SIZE per MI355X_MICROARCH.md section HBM; calibrated on BC1: "
                        "2*FETCH = texels read, WRITE = blocks written.")
    with open(os.path.join(dst, "pmc_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    by_wl = {}
    for wl in ALL_WORKLOADS:
        sqf = os.path.join(src, f"pmc_sq_{wl}.csv")
        if not os.path.exists(sqf):
            continue
        calls = abi_calls(src, f"pmc_sq_{wl}", overrides)
        row = {}
        for c in SQ_COUNTERS:
            v = per_call(sqf, c, calls)
            if v is not None:
                row[c] = v
        if row:
            row["abi_calls"] = calls
            row["per_kernel"] = per_call_by_kernel(sqf, "SQ_INSTS_VALU", calls)      # bench.py prices each kernel with its own instruction mix
            row["isolated"] = isolated_kernels(sqf, calls)
            by_wl[wl] = row
    if by_wl:
        by_wl["_source_sha256"] = stamp
        by_wl["_note"] = ("per C-ABI call at 4096x4096 (all kernels of the call; call count = `abi_calls` of the profiled bench line), rocprofv3 --pmc SQ pass of "
                          "tools/profile_gpu.sh; SQ cycle counters are in quad-cycles; bench.py turns SQ_INSTS_VALU into formats[*].valu; `isolated`: per kernel, "
                          "durations of the SERIALISED dispatches of this pass (they do not overlap, unlike the kernel-trace summary of a two-stream call)")
        with open(os.path.join(dst, f"{tag}_valu_by_workload.json"), "w") as fh:
            json.dump(by_wl, fh, indent=1)
        if "bc7_slow" in by_wl:                                # the headline's counters alone (older readers)
            out = {c: by_wl["bc7_slow"][c] for c in SQ_COUNTERS if c in by_wl["bc7_slow"]}
            out["_source_sha256"] = stamp
            out["_note"] = "per C-ABI call of bc7_slow at 4096x4096 (all kernels of the call); SQ cycle counters are in quad-cycles"
            with open(os.path.join(dst, f"{tag}_valu.json"), "w") as fh:
                json.dump(out, fh, indent=1)
    print(json.dumps({k: v for k, v in traffic.items() if not k.startswith("_")}, indent=1)[:2500])


if __name__ == "__main__":
    main()
