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


def per_call(path, counter):
    """Sum of `counter` over all kernels of one C-ABI call = total over the trace / number of calls, where the number
    of calls is the dispatch count of the LEAST frequent kernel name (every kernel of a call runs at least once per call; the bounded
    BC7 order launches bc7_scan_all twice)."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    if not rows:
        return None, 0
    by_kernel = collections.Counter(r["Kernel_Name"] for r in rows)
    calls = min(by_kernel.values())
    total = sum(float(r["Counter_Value"]) for r in rows)
    return total / calls, calls


def per_call_by_kernel(path, counter):
    """{kernel name: counter per C-ABI call} (same call count as per_call)."""
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    if not rows:
        return {}
    by_kernel = collections.Counter(r["Kernel_Name"] for r in rows)
    calls = min(by_kernel.values())
    out = collections.defaultdict(float)
    for r in rows:
        out[r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()] += float(r["Counter_Value"])
    return {k: v / calls for k, v in out.items()}


def main():
    tag = sys.argv[1]
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    try:
        stamp = open(os.path.join(src, "source_sha256.txt")).read().strip()
    except OSError:
        stamp = None
    for name in sorted(os.listdir(src)):
        if name.endswith((".csv", ".json")):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
            if stamp and name.startswith("kernel_stats"):            # csv files cannot carry the stamp themselves: a sidecar does
                with open(os.path.join(dst, f"{tag}_{name}.sha256"), "w") as fh:
                    fh.write(stamp + "\n")
    if stamp:
        with open(os.path.join(dst, f"{tag}_source_sha256.txt"), "w") as fh:
            fh.write(stamp + "\n")
    traffic = {}
    for wl in ("bc1", "bc3", "bc4", "bc5", "bc7_slow", "bc6h_slow"):
        f = os.path.join(src, f"pmc_{wl}_FETCH_SIZE.csv")
        w = os.path.join(src, f"pmc_{wl}_WRITE_SIZE.csv")
        if not (os.path.exists(f) and os.path.exists(w)):
            continue
        fetch_kib, n = per_call(f, "FETCH_SIZE")
        write_kib, _ = per_call(w, "WRITE_SIZE")
        if fetch_kib is None or write_kib is None:
            continue
        fetch = fetch_kib * 1024 * 2
        write = write_kib * 1024
        traffic[wl] = {"hbm_bytes_per_launch": int(fetch + write), "fetch_bytes_corrected_x2": int(fetch),
                       "write_bytes": int(write), "raw_FETCH_SIZE_KiB": fetch_kib, "raw_WRITE_SIZE_KiB": write_kib,
                       "calls_sampled": n, "source": f"profiles/{tag}_pmc_{wl}_FETCH_SIZE.csv, profiles/{tag}_pmc_{wl}_WRITE_SIZE.csv"}
    traffic["_source_sha256"] = stamp
    traffic["_note"] = ("rocprofv3 --pmc, one counter per pass (TCC slots); per C-ABI call = all kernels of the call; gfx950 x2 "
                        "correction applied to FETCH_SIZE per MI355X_MICROARCH.md section HBM; calibrated on BC1: 2*FETCH = "
                        "texels read, WRITE = blocks written.")
    with open(os.path.join(dst, "pmc_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    sq = os.path.join(src, "pmc_sq_bc7.csv")
    if os.path.exists(sq):
        out = {}
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"):
            v, n = per_call(sq, c)
            if v is not None:
                out[c] = v
        out["_source_sha256"] = stamp
        out["_note"] = "per C-ABI call of bc7_slow at 4096x4096 (all kernels of the call); SQ cycle counters are in quad-cycles"
        with open(os.path.join(dst, f"{tag}_valu.json"), "w") as fh:
            json.dump(out, fh, indent=1)
    by_wl = {}
    for wl in ("bc7_slow", "bc7_alpha_slow", "bc6h_slow", "bc1", "bc3", "bc4", "bc5"):
        sqf = os.path.join(src, f"pmc_sq_{wl}.csv")
        if not os.path.exists(sqf):
            continue
        row = {}
        for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAVES"):
            v, n = per_call(sqf, c)
            if v is not None:
                row[c] = v
        if row:
            row["per_kernel"] = per_call_by_kernel(sqf, "SQ_INSTS_VALU")      # bench.py prices each kernel with its own instruction mix
            by_wl[wl] = row
    if by_wl:
        by_wl["_source_sha256"] = stamp
        by_wl["_note"] = ("per C-ABI call at 4096x4096 (all kernels of the call), rocprofv3 --pmc SQ pass of tools/profile_gpu.sh; SQ cycle "
                          "counters are in quad-cycles; bench.py turns SQ_INSTS_VALU into formats[*].valu")
        with open(os.path.join(dst, f"{tag}_valu_by_workload.json"), "w") as fh:
            json.dump(by_wl, fh, indent=1)
    print(json.dumps(traffic, indent=1)[:1500])


if __name__ == "__main__":
    main()
