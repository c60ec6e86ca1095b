#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the BCn encode hot path on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one synthetic surface resident in HBM: CompressBlocks<fmt> called through
the drop-in C ABI with device pointers (kernel only; no PCIe in the timed region), plus -- when world_size > 1 --
the gather of the per-GPU output bands (the path's only exchange step).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload bc7_slow] [--size S] [--scaling weak|strong]

N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
(one process per GPU, RCCL).  Sharding is by block-row bands (itwBandForPart).  Rank 0 prints ONE JSON line.

  N = 1 (default)  BC7 `GetProfile_slow` on synthetic 4096x4096 RGBA8 -- BASELINE.json configs[2], the configuration
                   north_star's target is quoted on.  The same line carries `formats`: short measurements of
                   BC1 / BC3 / BC6H on the same size (configs[1], [3]) each with its own HBM roofline.
  N > 1 (default)  BASELINE.json configs[4]: ONE 16384x16384 RGBA8 surface (I5 = the 4096^2 surface tiled 4x4), BC7 slow,
                   strong-sharded: rank r encodes band r of the N bands, the output bands are all-gathered over xGMI
                   ("scaling": "strong"; --size 16384 --scaling strong at N = 1 runs the whole surface on one GPU).
                   A short weak-scaling figure (one 4096^2 band per rank) rides along as `weak_side`.
                   After the timed region the gathered image is CHECKED (`gather_verified`): every rank re-encodes its own
                   band and two other ranks' bands of the same seeded surface into fresh buffers and compares them byte for
                   byte with what the all-gather delivered; `per_rank_kernel_ms` shows load imbalance.  Collectives carry a
                   timeout (ITW_BENCH_DIST_TIMEOUT_S, default 300 s) and the whole run a watchdog (ITW_BENCH_WATCHDOG_S,
                   default 1500 s): a rank mismatch ends in an error, not in a hang.
  --scaling weak   rank r owns band r of a size x (size*N) surface: per-GPU work fixed.
  --host cpp|python|auto   who drives the N GPUs.  `cpp` = ONE process, itwCompressImageMultiGPUEx (include/itw_multigpu.h,
                   csrc/multigpu.hip): the C++ entry a host application calls -- one host thread per GPU, band r resident on
                   GPU r (tile-sharded input, BASELINE configs[4]), RCCL ncclSend/ncclRecv gather of the block stream to GPU 0,
                   every byte of the gathered image compared with a fresh single-GPU encode of its band.  `python` = the
                   torch.distributed job above.  auto (default): N > 1 runs BOTH -- the headline `value` is the C++ job's
                   (rank 0 runs it in a child process with a timeout while the other ranks wait at a CPU barrier; a child that
                   fails or hangs leaves the torch.distributed figure as the headline and says so in `cpp_host`), the
                   torch.distributed job rides along as `python_side`.  N = 1: `formats["multigpu_cpp@8virtual_16384"]` times
                   the same C++ entry with 8 ranks sharing the one device against the single 16384^2 call (dispatch overhead).

cpu_baseline: the scalar C oracle (oracle/, test infrastructure) timed on this box's host cores on a bounded sample,
rank 0, N=1 only.  It is a *port* (scalar restatement), not ISPC SIMD code: the reference cannot be built here.
"""
import argparse
import datetime
import faulthandler
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "intel-texture-works-plugin_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
VALU_PEAK_TOPS = 78.6          # 256 CU x 4 SIMD x 32 lanes x 2.4 GHz, separate mul/add (contract=off => no FMA credit)
VALU_PEAK_FMA_TFLOPS = 157.3   # the guide's fp32 vector peak: the same issue rate with FMA / packed-fp32 credit (2 flops per lane-op)

# algorithmic bytes per 4x4 block: texels read + block written
ALG_BYTES = {"bc1": 64 + 8, "bc3": 64 + 16, "bc7": 64 + 16, "bc6h": 128 + 16, "bc4": 64 + 8, "bc5": 64 + 16}

WORKLOADS = {
    "bc1": ("bc1", None), "bc3": ("bc3", None), "bc4": ("bc4", None), "bc5": ("bc5", None),
    "bc7_ultrafast": ("bc7", "ultrafast"), "bc7_veryfast": ("bc7", "veryfast"), "bc7_fast": ("bc7", "fast"),
    "bc7_basic": ("bc7", "basic"), "bc7_slow": ("bc7", "slow"),
    "bc7_alpha_veryfast": ("bc7", "alpha_veryfast"), "bc7_alpha_basic": ("bc7", "alpha_basic"), "bc7_alpha_slow": ("bc7", "alpha_slow"),
    "bc6h_fast": ("bc6h", "fast"), "bc6h_basic": ("bc6h", "basic"), "bc6h_slow": ("bc6h", "slow"),
}


_SURFACES = {}


def make_surface(fmt, size, rank):
    """Seeded synthetic surface (cached: the N > 1 jobs cut several bands from the same base)."""
    key = (fmt == "bc6h", size, rank)
    if key not in _SURFACES:
        if len(_SURFACES) > 4:
            _SURFACES.clear()
        _SURFACES[key] = _make_surface(fmt, size, rank)
    return _SURFACES[key]


def _make_surface(fmt, size, rank):
    from itw_amd import surfaces
    if fmt == "bc6h":
        return surfaces.hdr_smooth(size, size, seed=surfaces.SEED + 3 + 100 * rank)
    return surfaces.ldr_smooth(size, size, seed=surfaces.SEED + 100 * rank)


def plan(scaling, size, world, rank, fmt, piece=0, pieces=1):
    """Geometry of the job and of this rank's share.  Pure (no GPU): tests/test_sharding_gloo.py checks it.
    weak  : surface = size wide x (size * world) tall, rank r owns the r-th size x size band.
    strong: surface = size x size (BASELINE configs[4] at size 16384), rank r owns block rows [R*r/N, R*(r+1)/N) -- or, with
            pieces = K > 1 (the content-aware partition, include/itw_multigpu.h), its `piece`-th of K interleaved sub-bands: sub-band
            piece * N + r of K * N.
    Returns dict(width, height, y0, rows, band_off, band_bytes, total_bytes)."""
    from itw_amd import shard, abi
    width, height = (size, size * world) if scaling == "weak" else (size, size)
    y0, rows, off, nbytes = shard.sub_band_of(width, height, fmt, rank, world, piece, pieces)
    total = (width // 4) * (height // 4) * abi.BYTES_PER_BLOCK[fmt]
    return {"width": width, "height": height, "y0": y0, "rows": rows, "band_off": off, "band_bytes": nbytes, "total_bytes": total}


def pieces_for(size, world, interleave=None):
    """K of the strong-sharded job: the requested sub-bands per rank (default 4) halved until the block rows divide by K * N and every
    sub-band keeps 16 block rows (equal sub-bands: the all-gather is in place)."""
    K = interleave if interleave else 4
    while K > 1 and ((size // 4) % (K * world) != 0 or (size // 4) // (K * world) < 16):
        K //= 2
    return max(1, K)


def dry_run_plan(size, world, fmt, interleave=None):
    """--dry-run-plan N: what `--gpus N` WILL do on the strong-sharded surface, without a GPU -- every rank's sub-bands (texel rows, input
    bytes, offset and length in the block stream), the K in-place all-gathers with their message sizes, and the C++ job's sends to the
    owner.  tests/test_sharding_gloo.py checks it against itwBandForPart; the first hardware run can be compared line by line."""
    from itw_amd import abi
    K = pieces_for(size, world, interleave)
    bpb = abi.BYTES_PER_BLOCK[fmt]
    texel = 8 if fmt == "bc6h" else 4
    ranks = []
    for r in range(world):
        subs = []
        for k in range(K):
            g = plan("strong", size, world, r, fmt, k, K)
            subs.append({"sub_band": k * world + r, "piece": k, "first_texel_row": g["y0"], "texel_rows": g["rows"],
                         "input_bytes": g["rows"] * size * texel, "out_offset": g["band_off"], "out_bytes": g["band_bytes"]})
        ranks.append({"rank": r, "device": r, "sub_bands": subs, "texel_rows": sum(x["texel_rows"] for x in subs),
                      "input_bytes": sum(x["input_bytes"] for x in subs), "out_bytes": sum(x["out_bytes"] for x in subs)})
    piece_bytes = ranks[0]["sub_bands"][0]["out_bytes"]
    total = (size // 4) ** 2 * bpb
    return {"plan": f"{fmt} on ONE {size}x{size} surface, strong-sharded over {world} ranks", "size": size, "ranks": world, "sub_bands_per_rank": K,
            "bytes_per_block": bpb, "total_out_bytes": total, "per_rank": ranks,
            "python_job_collectives": [{"group": k, "op": "all_gather_into_tensor (in place)", "send_bytes_per_rank": piece_bytes,
                                        "recv_bytes_per_rank": piece_bytes * (world - 1), "buffer_offset": k * world * piece_bytes,
                                        "buffer_bytes": world * piece_bytes, "overlaps": f"encode of piece {k + 1}" if k + 1 < K else "next step's first encode"}
                                       for k in range(K)],
            "cpp_job_gather": {"op": "grouped ncclSend / ncclRecv to the rank that owns the output (rank 0)", "groups": K,
                               "send_bytes_per_rank_per_group": piece_bytes, "owner_recv_bytes_total": total - K * piece_bytes},
            "xgmi_bytes_per_link_python": piece_bytes * K, "note": "all-gather: every rank sends its K pieces to each of the N-1 peers, one xGMI link per peer"}


def make_band(fmt, scaling, size, geo, rank):
    """Texels of this rank's band.  weak: an independent size x size surface per rank.  strong: rows [y0, y0+rows) of
    I5 = the 4096^2 synthetic surface tiled up to size x size -- every rank derives them from the same
    seeded base, so the job is one well-defined image and no scatter is timed (synthetic data)."""
    from itw_amd import surfaces
    if scaling == "weak" or size <= 4096:
        img = make_surface(fmt, size, rank if scaling == "weak" else 0)
        return np.ascontiguousarray(img[geo["y0"]:geo["y0"] + geo["rows"]]) if scaling == "strong" else img
    base = make_surface(fmt, 4096, 0)
    rows = (np.arange(geo["y0"], geo["y0"] + geo["rows"]) % 4096)
    cols = (np.arange(geo["width"]) % 4096)
    return np.ascontiguousarray(base[rows][:, cols])


def fake_blocks(fmt, img):
    """Stand-in "encoder" of the CPU control-flow test: every block's stream bytes = the bytes of its top-left texel, tiled.
    A deterministic function of the texels, so the gather verification has something real to compare."""
    from itw_amd import abi
    bpb = abi.BYTES_PER_BLOCK[fmt]
    tl = np.ascontiguousarray(img[0::4, 0::4]).reshape(-1, img.shape[2] * img.itemsize).view(np.uint8)
    reps = -(-bpb // tl.shape[1])
    return np.ascontiguousarray(np.tile(tl, (1, reps))[:, :bpb]).reshape(-1)


ABI_CALLS = [0]        # C-ABI encode calls this process has made: tools/summarize_profiles.py divides a profiled run's counters by it


def _compress(itw, fmt, d_img, prof, out):
    ABI_CALLS[0] += 1
    return itw.compress(fmt, d_img, prof, out=out)


def make_encoder(itw, fmt, prof, img, dev):
    """(encode_into(out_band), device texels or None) for one band of texels."""
    if FAKE:
        blocks = torch.from_numpy(fake_blocks(fmt, img))
        return (lambda out: out.copy_(blocks)), None
    d_img = torch.from_numpy(img).to(dev)
    return (lambda out: _compress(itw, fmt, d_img, prof, out)), d_img


def verify_gather(itw, dist, full, scaling, size, world, rank, fmt, prof, dev, pieces=1):
    """Correctness of the N > 1 job, after the timed region: `full` is the whole-image stream the last step's all-gather
    left on this rank.  This rank re-encodes (a) its own band -- it must have survived the in-place gather -- and (b) the
    bands of ranks rank+1 and rank+N/2 (so every band is checked by up to three different ranks) from the same seeded
    surface into FRESH buffers and compares them with the gathered bytes.  Returns the job-wide verdict (all ranks agree)."""
    targets = []
    for r in (rank, (rank + 1) % world, (rank + world // 2) % world):
        if r not in targets:
            targets.append(r)
    bad_bytes = 0
    for r in targets:
        for k in range(pieces):                              # (pieces > 1: every sub-band of that rank)
            g = plan(scaling, size, world, r, fmt, k, pieces)
            enc, keep = make_encoder(itw, fmt, prof, make_band(fmt, scaling, size, g, r), dev)
            ref = torch.empty(g["band_bytes"], dtype=torch.uint8, device=dev)
            enc(ref)
            _sync()
            bad_bytes += int((ref != full[g["band_off"]:g["band_off"] + g["band_bytes"]]).sum().item())
            del ref, keep, enc
    t = torch.tensor([bad_bytes, len(targets) * pieces], dtype=torch.int64, device=dev)
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return {"gather_verified": int(t[0].item()) == 0, "mismatching_bytes": int(t[0].item()), "band_checks": int(t[1].item()),
            "how": "every rank re-encoded its own band and the bands of ranks r+1 and r+N/2 of the same seeded surface into "
                   "fresh buffers and compared them byte for byte with the all-gathered stream of the last timed step"}


# ITW_BENCH_CONTROL_FLOW_TEST=1: no GPU, gloo instead of RCCL, the encode replaced by a memset.  NOT a measurement: it exists
# so that tests/test_sharding_gloo.py can execute the N > 1 control flow of this file (plan, bands, pipelined gather, max over
# ranks, weak side figure, the JSON line) on CPU before the driver runs it on eight GPUs.
FAKE = os.environ.get("ITW_BENCH_CONTROL_FLOW_TEST") == "1"


def _sync():
    if not FAKE:
        torch.cuda.synchronize()


def time_kernel(itw, fmt, prof, d_img, d_out, steps, warmup, back_to_back=False):
    """Average launch duration (ms) from HIP events recorded on the stream the kernel runs on.  back_to_back: ONE event pair
    around all `steps` launches (for kernels of a few tens of microseconds, where an event pair per launch adds 2-3 us of
    its own); returns (average, average)."""
    if FAKE:
        return 1.0, 1.0
    for _ in range(warmup):
        _compress(itw, fmt, d_img, prof, d_out)
    torch.cuda.synchronize()
    if back_to_back:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(steps):
            _compress(itw, fmt, d_img, prof, d_out)
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / steps
        return float(t), float(t)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    for a, b in evs:
        a.record()
        _compress(itw, fmt, d_img, prof, d_out)
        b.record()
    torch.cuda.synchronize()
    ts = [a.elapsed_time(b) for a, b in evs]
    return float(np.mean(ts)), float(np.min(ts))


def _cpu_encoder():
    """(encode_mt, kind, description) of the CPU leg: the reference's own kernel.ispc built as a scalar program
    (oracle/_ref/libispc_texcomp_ref_full.so, oracle/ref_build/ispc_as_cpp/: kind "reference") when that prebuilt library
    is there, else the oracle's C restatement (kind "port").  Same algorithm, same bytes, both scalar -- neither is ISPC SIMD."""
    from oracle import pyoracle            # checker / baseline leg only
    pyoracle.build()
    try:
        from oracle import pyref
        if pyref.available():
            pyref.lib()
            return pyref.encode_mt, "reference", "the reference's kernel.ispc compiled as ONE scalar program instance (clang -O2, no ispc: not ISPC SIMD)"
    except OSError:
        pass
    return pyoracle.encode_mt, "port", "scalar C oracle (not ISPC SIMD)"


def cpu_baseline(fmt, prof, img, budget_s=15.0):
    """CPU path on host cores, bounded sample of the same surface: grow the band until ~budget_s of CPU work."""
    from oracle import pyoracle            # checker / baseline leg only
    encode_mt, kind, what = _cpu_encoder()
    cores = pyoracle.usable_cores()           # honours the cgroup CPU quota of the box
    h, w = img.shape[:2]
    rows = min(h, max(4 * cores, 16))
    encode_mt(fmt, img[:rows], prof, threads=cores)          # warm the thread pool / caches
    t0 = time.perf_counter()
    encode_mt(fmt, img[:rows], prof, threads=cores)
    dt = time.perf_counter() - t0
    if dt < budget_s / 4 and rows < h:                                 # grow the band towards the budget
        rows = int(min(h, max(rows, rows * (budget_s * 0.8) / max(dt, 1e-3)))) // 4 * 4
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < min(budget_s, 5.0) or reps == 0:                        # short formats: repeat the sample
        encode_mt(fmt, img[:rows], prof, threads=cores)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > budget_s:
            break
    dt /= reps
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(rows * w / dt / 1e6, 3), "unit": "Mpixels/s", "cores": cores, "kind": kind,
            "sample": f"first {rows} of {h} texel rows of the same {w}x{h} surface, {reps} x {dt:.3f} s, "
                      f"{what}, {cores} threads (= usable cores: min of cpu_count "
                      f"{os.cpu_count()}, affinity, cgroup quota), reference band rule; cpu: {model}"}


def cpu_baseline_pair(fmt, prof, img, budget_s=2.0):
    """Mpixels/s of the CPU path (see _cpu_encoder), 1 thread and all usable threads (side formats).  A leg is a sample grown towards
    ~budget_s of work (up to the whole surface) and REPEATED until the budget is spent; the best repetition counts.  (Round 4 timed one
    call on a 1024-row surface: 20 ms of BC1 work, of which starting 16 threads was most -- 16 threads came out slower than 1.  The
    caller now hands the fast formats their whole 4096^2 surface; the reference's band rule is encode_mt's: win32Threads.cpp:217-231.)"""
    from oracle import pyoracle            # checker / baseline leg only
    encode_mt, kind, what = _cpu_encoder()
    cores = pyoracle.usable_cores()
    h, w = img.shape[:2]
    out = {"unit": "Mpixels/s", "kind": kind, "cores": cores}
    for label, n in (("threads_1", 1), ("threads_all", cores)):
        rows = min(h, max(4 * n, 64))
        encode_mt(fmt, img[:rows], prof, threads=n)                    # warm-up (page faults of the output, thread start)
        t0 = time.perf_counter()
        encode_mt(fmt, img[:rows], prof, threads=n)
        dt = max(time.perf_counter() - t0, 1e-5)
        rows = int(min(h, max(rows, rows * (budget_s / 3) / dt))) // 4 * 4    # about a third of the budget per repetition, if the surface has it
        best, reps, spent = 1e30, 0, 0.0
        while reps < 3 or (spent < budget_s and reps < 200):
            t0 = time.perf_counter()
            encode_mt(fmt, img[:rows], prof, threads=n)
            dt = time.perf_counter() - t0
            best = min(best, dt); spent += dt; reps += 1
        out[label] = round(rows * w / best / 1e6, 3)
        out[label + "_sample"] = f"{rows} rows x {reps} repetitions, best {best * 1e3:.2f} ms"
    out["sample"] = f"up to {h} rows of a {w}-wide synthetic surface, ~{budget_s:.0f} s per leg, best repetition, {what}"
    return out


def _latest_profile(suffix):
    """(path, parsed JSON) of the latest committed profiles/*<suffix>, or (None, None)."""
    try:
        names = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith(suffix))
        if not names:
            return None, None
        path = os.path.join(ROOT, "profiles", names[-1])
        with open(path) as f:
            return path, json.load(f)
    except (OSError, ValueError):
        return None, None


def pmc_valu(workload):
    """VALU wave-instructions per C-ABI call at 4096^2 from the latest committed rocprofv3 SQ pass (profiles/*_valu_by_workload.json,
    tools/profile_gpu.sh): (instructions per call, {kernel name: instructions}, source stamp of the pass)."""
    path, j = _latest_profile("_valu_by_workload.json")
    if not j or workload not in j:
        return None, None, None
    return j[workload].get("SQ_INSTS_VALU"), j[workload].get("per_kernel"), j.get("_source_sha256")


def isolated_kernels(workload, cur_sha):
    """rocprof_kernels.isolated: per kernel of one C-ABI call, from the SERIALISED dispatches of the committed SQ counter pass
    (profiles/*_valu_by_workload.json `isolated`, tools/summarize_profiles.py): ms, VALU wave-instructions, the lane-op fraction of the
    78.6 T peak and the issue fraction (cycles per instruction of that kernel from the weighted ISA model).  These durations do not
    overlap -- the kernel-trace summary's do since a call is two bands on two streams -- so their sum is >= the call's time."""
    path, j = _latest_profile("_valu_by_workload.json")
    if not j or workload not in j or j.get("_source_sha256") != cur_sha or "isolated" not in j[workload]:
        return None
    _, model = _latest_profile("_issue_model.json")
    by_name = (model or {}).get("by_rocprof_name", {}) if (model or {}).get("_source_sha256") == cur_sha else {}
    rows = {}
    for k, v in j[workload]["isolated"].items():
        if v["ms"] <= 0:
            continue
        lane = v["wave_valu"] * 64 / (v["ms"] * 1e-3) / 1e12
        row = {"ms": round(v["ms"], 4), "dispatches_per_call": v["dispatches"], "wave_valu": int(v["wave_valu"]),
               "lane_op_frac": round(lane / VALU_PEAK_TOPS, 4), "vgprs": v.get("vgprs"), "lds_bytes_per_workgroup": v.get("lds_bytes")}
        hit = [m for n, m in by_name.items() if n.split("(")[0].strip() in k]
        if hit:
            row["issue_frac"] = round(v["wave_valu"] * hit[0]["cycles_per_instruction"] / (v["ms"] * 1e-3 * 2.4e9 * 1024), 4)
        rows[k] = row
    return {"kernels": rows, "sum_ms": round(sum(r["ms"] for r in rows.values()), 4),
            "source": os.path.relpath(path, ROOT) + " (rocprofv3 --pmc SQ pass: dispatches serialised, durations include the counter readout)"}


def issue_block(per_kernel, insts, kernel_ms, cur_sha):
    """The roofline that binds, in the unit that binds: issue cycles the executed VALU instructions NEED --
    every kernel's measured wave-instruction count x its cycles per instruction from the trip-count-weighted ISA model
    (profiles/*_issue_model.json, tools/isa_weighted.py: 2 cycles for the plain VOP2 forms, 4 for the rest, 8 for rcp / sqrt /
    64-bit multiplies) -- against the cycles the chip HAS during the call: time x 2.4 GHz x 1024 SIMDs."""
    path, model = _latest_profile("_issue_model.json")
    if not model or not insts:
        return None
    by_name = model.get("by_rocprof_name", {})
    need, used, missing = 0.0, {}, []
    for kname, n in (per_kernel or {}).items():
        hit = [v for k, v in by_name.items() if k.split("(")[0].strip() in kname]
        if hit:
            need += n * hit[0]["cycles_per_instruction"]
            used[kname[:60]] = hit[0]["cycles_per_instruction"]
        else:
            missing.append(kname[:60])
    if not used:
        return None
    if missing:                                               # kernels without a model: the call's average mix
        cpi = need / sum(n for k, n in per_kernel.items() if k[:60] in used)
        need += sum(per_kernel[k] for k in per_kernel if k[:60] in missing) * cpi
    avail = kernel_ms * 1e-3 * 2.4e9 * 1024
    return {"cycles_needed": int(need), "cycles_available": int(avail), "frac": round(need / avail, 4),
            "cycles_per_instruction": used, "model": os.path.relpath(path, ROOT),
            "model_matches_source": model.get("_source_sha256") == cur_sha,
            "unit": "SIMD issue cycles per C-ABI call (1 024 SIMDs x 2.4 GHz nominal; a wave64 VALU instruction holds its SIMD for 2, 4 or 8 cycles)",
            "note": "1 - frac = issue slots the call leaves empty (dependency stalls, s_nop, LDS / memory waits, launch ramp and tail) -- at the "
                    "NOMINAL clock: measured shader clocks under these kernels (s_memtime against the 100 MHz clock, profiles/history/r05/r04_clock_probe.txt) "
                    "are 2.31 GHz in the BC7 scans and 2.04 GHz in the BC1 / BC3 kernels, i.e. frac / 0.96 and frac / 0.85 of the cycles the chip really had"}


def valu_block(insts, kernel_ms):
    """The roofline that binds: executed VALU wave-instructions x 64 lanes / kernel time against the 2-cycle-issue peak."""
    lane_ops = insts * 64 / (kernel_ms * 1e-3) / 1e12
    return {"achieved": round(lane_ops, 2), "peak": VALU_PEAK_TOPS, "unit": "T lane-ops/s", "frac": round(lane_ops / VALU_PEAK_TOPS, 4),
            "wave_instructions_per_call": int(insts)}


def op_counts(workload):
    """Measured fp32 operation counts per block of the reference algorithm (profiles/op_counts.json: the CPU oracle under
    ptrace single-stepping, tools/opcount.py) -- the ALGORITHMIC work, as opposed to the instructions this GPU
    implementation happens to execute."""
    try:
        with open(os.path.join(ROOT, "profiles", "op_counts.json")) as f:
            j = json.load(f)
        return j["workloads"][workload]["per_block"], j.get("method")
    except (OSError, ValueError, KeyError):
        return None, None


def rocprof_kernels(fmt):
    """Per-kernel average duration (ms) of this format's kernels from the latest committed `rocprofv3 --kernel-trace --stats`
    summary (profiles/*_kernel_stats.csv), so the live HIP-event time of the call can be checked against the profiler's."""
    import csv
    import re
    pat = {"bc7": "bc7_", "bc6h": "bc6h_kernel", "bc1": "bc13_kernel<false", "bc3": "bc13_kernel<true",
           "bc4": "bc45_kernel<1", "bc5": "bc45_kernel<2"}[fmt]
    try:
        suffix = "_kernel_stats.csv" if fmt == "bc7" else "_kernel_stats_all_formats.csv"   # headline-only pass / all formats
        names = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith(suffix)) \
            or sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith("_kernel_stats.csv"))
        if not names:
            return None
        out, totals, calls = {}, {}, {}
        with open(os.path.join(ROOT, "profiles", names[-1])) as f:
            for r in csv.DictReader(f):
                if pat in r["Name"]:
                    m = re.search(r"(bc\w+<[^>]*>)", r["Name"])
                    key = m.group(1) if m else re.sub(r"\(.*", "", r["Name"]).replace("itw::", "")[:48]
                    out[key] = round(float(r["AverageNs"]) / 1e6, 4)
                    totals[key] = float(r["TotalDurationNs"]) / 1e6
                    calls[key] = int(r["Calls"])
        stamp = None
        try:
            stamp = open(os.path.join(ROOT, "profiles", names[-1] + ".sha256")).read().strip()
        except OSError:
            pass
        if not out:
            return None
        # C-ABI calls in the trace = dispatches of the least frequent kernel name (BC7 `slow`: the pilot's estimate runs once per call).  Since
        # round 5 a call launches BOTH continuations of each band behind the pilot and the one not chosen returns at once, so a kernel
        # name's AVERAGE mixes real and empty launches: the per-call totals are what to compare with the live time (the two bands run on
        # two streams, so the totals add up to more than the call's duration).
        ncalls = min(calls.values())
        return {"source": "profiles/" + names[-1], "avg_ms": out, "launches_per_call": {k: round(v / ncalls, 2) for k, v in calls.items()},
                "per_call_ms": {k: round(v / ncalls, 4) for k, v in totals.items()}, "source_sha256": stamp}
    except (OSError, ValueError, KeyError):
        return None


def pmc_traffic(workload, cur_sha=None):
    """HBM bytes per launch from a committed rocprofv3 --pmc pass (profiles/pmc_traffic.json) -- None when the pass was taken on other
    kernel sources than the tree this runs from (its stamp, written by tools/profile_gpu.sh, differs)."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            j = json.load(f)
        if cur_sha is not None and j.get("_source_sha256") != cur_sha:
            return None
        return j.get(workload, {}).get("hbm_bytes_per_launch")
    except (OSError, ValueError):
        return None


def _all_device_sync():
    for i in range(torch.cuda.device_count()):
        torch.cuda.synchronize(i)


def cpp_job(itw, fmt, prof, size, ranks, steps, warmup, scatter_steps=0):
    """BASELINE configs[4] through the C++ host path: ONE process, itwCompressImageMultiGPUEx, band r of the size^2 surface resident
    on device r % device_count (each band derived from the same seeded base as the torch.distributed job's), block stream gathered
    to device 0.  Every call is synchronous (all rank streams drained before it returns), so the timed region is K calls between
    two all-device synchronisations.  Afterwards EVERY band of the gathered image is compared with a fresh single-GPU encode."""
    ndev = torch.cuda.device_count()
    bands, geos = [], []
    K = itw.lib().itwMultiGpuPieces(size, ranks, 0)          # sub-bands per rank the library uses for this geometry (itw_multigpu.h; default 4)
    for j in range(K * ranks):                               # sub-band j of K * ranks, resident on the device of rank j % ranks
        g = plan("strong", size, ranks, j % ranks, fmt, j // ranks, K)
        geos.append(g)
        bands.append(torch.from_numpy(make_band(fmt, "strong", size, g, j % ranks)).to(f"cuda:{(j % ranks) % ndev}"))
    out = torch.zeros(geos[0]["total_bytes"], dtype=torch.uint8, device="cuda:0")
    st = itw.MultiGpuStats()
    for _ in range(warmup):
        itw.compress_image_multigpu(fmt, (size, size), prof, ranks=ranks, bands=bands, out=out, stats=st)
    first = st.as_dict() if warmup else None               # the first call of the process: communicator set-up, first connections
    _all_device_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        itw.compress_image_multigpu(fmt, (size, size), prof, ranks=ranks, bands=bands, out=out, stats=st)
    _all_device_sync()
    elapsed = time.perf_counter() - t0
    last = st.as_dict()
    bad = 0
    for r in range(K * ranks):
        ref = itw.compress(fmt, bands[r], prof)
        torch.cuda.synchronize(ref.device)
        g = geos[r]
        bad += int((ref.to("cuda:0") != out[g["band_off"]:g["band_off"] + g["band_bytes"]]).sum().item())
        del ref
    res = {"host": "cpp: one process, itwCompressImageMultiGPUEx, one host thread per rank", "elapsed_s": elapsed, "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 4), "value": round(size * size * steps / elapsed / 1e6, 2), "unit": "Mpixels/s",
           "gather_verified": bad == 0, "mismatching_bytes": bad, "band_checks": K * ranks, "sub_bands_per_rank": K,
           "how": "every band of the gathered stream on GPU 0 compared byte for byte with a fresh single-GPU encode of the same texels",
           "ranks": ranks, "devices": ndev, "transport": last["transport"], "transport_note": last["transport_note"],
           "ranks_seen_by_rccl": last["rccl_ranks"], "peer_links": last["peer_links"], "stats_last_call": last,
           "first_call_wall_ms": first["wall_ms"] if first else None}
    if scatter_steps > 0:
        # the other way a C++ host holds its texels: the WHOLE surface resident on GPU 0, scattered to the ranks by peer copies
        # inside the call (the second half-band's copy under the first's encode)
        whole = torch.cat([b.to("cuda:0") for b in bands], dim=0)          # (sub-bands are in surface order)
        o2 = torch.zeros_like(out)
        itw.compress_image_multigpu(fmt, whole, prof, ranks=ranks, out=o2, stats=st)
        _all_device_sync()
        t0 = time.perf_counter()
        for _ in range(scatter_steps):
            itw.compress_image_multigpu(fmt, whole, prof, ranks=ranks, out=o2, stats=st)
        _all_device_sync()
        e2 = time.perf_counter() - t0
        res["scatter_from_gpu0"] = {"ms_per_step": round(e2 / scatter_steps * 1e3, 4), "value": round(size * size * scatter_steps / e2 / 1e6, 2),
                                    "unit": "Mpixels/s", "steps": scatter_steps, "identical_to_resident_bands_result": bool(torch.equal(o2, out)),
                                    "stats_last_call": st.as_dict()}
        del whole, o2
    return res


def cpp_worker(args):
    """Child process of rank 0 (N > 1): runs cpp_job over all visible GPUs and prints its result as one JSON line."""
    fmt, prof = WORKLOADS[args.workload]
    if FAKE:                                               # control-flow test on CPU (tests/test_sharding_gloo.py): a canned account
        if os.environ.get("ITW_BENCH_CPP_WORKER_CRASH"):
            raise SystemExit(3)
        print(json.dumps({"host": "cpp (CONTROL-FLOW TEST, canned)", "elapsed_s": 0.002 * args.steps, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": 2.0, "value": round(args.size * args.size / 2e-3 / 1e6, 2), "unit": "Mpixels/s",
                          "gather_verified": os.environ.get("ITW_BENCH_FAKE_CPP") != "bad", "mismatching_bytes": 0, "band_checks": args.gpus,
                          "how": "canned", "ranks": args.gpus, "devices": args.gpus, "transport": "rccl", "transport_note": "", "ranks_seen_by_rccl": args.gpus,
                          "peer_links": args.gpus * (args.gpus - 1), "stats_last_call": None, "first_call_wall_ms": None}), flush=True)
        return
    import itw_amd
    itw_amd.lib()
    assert torch.cuda.is_available(), "cpp worker: no GPU"
    if torch.cuda.device_count() < args.gpus and os.environ.get("ITW_BENCH_CPP_SHARE_DEVICES") != "1":
        # (ITW_BENCH_CPP_SHARE_DEVICES=1: ranks share devices, rank r on device r % count -- how the job is exercised on a 1-GPU box)
        raise SystemExit(f"cpp worker: {torch.cuda.device_count()} visible devices for {args.gpus} ranks")
    faulthandler.dump_traceback_later(int(os.environ.get("ITW_BENCH_CPP_TIMEOUT_S", "600")), exit=True)
    res = cpp_job(itw_amd, fmt, prof, args.size, args.gpus, args.steps, args.warmup, scatter_steps=min(3, args.steps))
    print(json.dumps(res), flush=True)


def run_cpp_child(args, world, size, steps, warmup):
    """Rank 0 of an N > 1 launch: the C++ job in a child process (its own HIP contexts on all N GPUs; torchrun's variables
    removed), bounded by a timeout.  Returns its result dict, or {"error": ...}."""
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "GROUP_WORLD_SIZE", "ROLE_RANK", "ROLE_NAME",
                        "ROLE_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "OMP_NUM_THREADS") and not k.startswith("TORCHELASTIC_")}
    cmd = [sys.executable, os.path.abspath(__file__), "--cpp-worker", "--gpus", str(world), "--steps", str(steps), "--warmup", str(warmup),
           "--workload", args.workload, "--size", str(size)]
    tmo = int(os.environ.get("ITW_BENCH_CPP_TIMEOUT_S", "600"))
    try:
        p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=tmo + 30)
    except subprocess.TimeoutExpired:
        return {"error": f"the C++ multi-GPU job did not finish within {tmo + 30} s and was killed"}
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not lines:
        return {"error": f"the C++ multi-GPU job exited with {p.returncode}", "stderr_tail": p.stderr[-1500:]}
    try:
        return json.loads(lines[-1])
    except ValueError as e:
        return {"error": f"unreadable result of the C++ multi-GPU job: {e}"}


def sliced_side(itw_amd, size, make_surface):
    """formats["sliced@64"]: the plugin's slice loop (IntelPlugin.cpp:851-879: 0x40000-pixel slices, SetProgress between them) through
    itwCompressImageSliced with a progress callback installed, HOST pointers (pageable memory), per trampoline the plugin selects (+ `slow`):
    the pipeline (windows of slices in flight, csrc/abi.hip compress_sliced) beside the literal loop and ONE CompressImageST call."""
    import ctypes as C
    L = itw_amd.lib()
    out = {"what": "4096^2, 64 slices of 0x40000 px, host pointers, progress callback per slice; Mpixels/s = best of 5 synchronous calls after a warm-up",
           "slices": max(1, size * size // 0x40000)}
    calls = []
    cb = itw_amd.PROGRESS_FUNC(lambda i, n, u: calls.append(i) or True)
    for fmt, prof in (("bc1", None), ("bc3", None), ("bc7", "veryfast"), ("bc7", "basic"), ("bc7", "alpha_veryfast"), ("bc7", "alpha_basic"),
                      ("bc7", "slow"), ("bc6h", "fast"), ("bc6h", "slow")):
        name = fmt + ("_" + prof if prof else "")
        try:
            img = make_surface(fmt, size, 0)
            h, w = img.shape[:2]
            nbytes = itw_amd.block_count(fmt, w, h) * itw_amd.BYTES_PER_BLOCK[fmt]
            got, want = np.zeros(nbytes, dtype=np.uint8), np.zeros(nbytes, dtype=np.uint8)
            surf = itw_amd.RgbaSurface(img.ctypes.data, w, h, img.strides[0])
            fn, code = itw_amd.image_func(fmt, prof), itw_amd.DXGI_FORMAT[fmt]
            pitch = itw_amd.block_count(fmt, w, 4) * itw_amd.BYTES_PER_BLOCK[fmt]

            def best(f):
                f()
                ts = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    f()
                    ts.append(time.perf_counter() - t0)
                return min(ts) * 1e3
            one = best(lambda: L.CompressImageST(C.byref(surf), want.ctypes.data, fn, code))
            row = {"one_call_ms": round(one, 3), "one_call_Mpixels/s": round(w * h / one / 1e3, 1)}
            for key, W in (("literal_loop", -1), ("pipeline", 0)):
                L.itwSetSliceWindow(W)
                got[:] = 0
                del calls[:]
                ms = best(lambda: L.itwCompressImageSliced(C.byref(surf), got.ctypes.data, pitch, fn, code, False, 0, C.cast(cb, C.c_void_p), None))
                row[key + "_ms"] = round(ms, 3)
                row[key + "_Mpixels/s"] = round(w * h / ms / 1e3, 1)
                row[key + "_bytes_equal_one_call"] = bool(np.array_equal(got, want))
                if W == 0:
                    st7 = itw_amd.bc7_profile(prof) if fmt == "bc7" else None
                    row["window_slices"] = int(L.itwSliceWindowFor(code, C.cast(C.byref(st7), C.c_void_p) if st7 is not None else None, w, h, 0))
                    row["progress_calls_per_run"] = len(calls) // 6
            L.itwSetSliceWindow(0)
            row["pipeline_over_one_call"] = round(one / row["pipeline_ms"], 3)
            out[name] = row
        except Exception as e:
            itw_amd.lib().itwSetSliceWindow(0)
            out[name] = {"error": repr(e)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="bc7_slow", choices=sorted(WORKLOADS))
    ap.add_argument("--size", type=int, default=None, help="surface edge; default 4096 (N = 1 / weak) or 16384 (strong, N > 1)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default=None,
                    help="default: strong when N > 1 (BASELINE configs[4]: one 16384^2 surface sharded over the ranks), else weak")
    ap.add_argument("--interleave", type=int, default=None, help="K sub-bands per rank of the strong-sharded job (content-aware partition, "
                    "include/itw_multigpu.h); default 4 where the surface's block rows divide by K * N, else 1")
    ap.add_argument("--no-formats", action="store_true", help="skip the side measurements of the other formats")
    ap.add_argument("--no-16k", action="store_true", help="skip the 16384^2 BC1/BC3 side figures (tools/profile_gpu.sh: keeps the "
                    "per-kernel-name averages of rocprofv3 --stats those of the 4096^2 launches)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--host", choices=["auto", "cpp", "python"], default="auto",
                    help="N > 1: who drives the GPUs (see the module docstring); auto = both, C++ job as the headline")
    ap.add_argument("--cpp-worker", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dry-run-plan", type=int, default=0, metavar="N", help="print what --gpus N will do on the strong-sharded surface (every rank's "
                    "sub-bands, offsets, message sizes) as one JSON line and exit; needs no GPU")
    args = ap.parse_args()
    if args.dry_run_plan:
        print(json.dumps(dry_run_plan(args.size or 16384, args.dry_run_plan, WORKLOADS[args.workload][0], args.interleave)), flush=True)
        return
    if args.cpp_worker:
        args.size = args.size or 16384
        args.steps = args.steps if args.steps is not None else 10
        args.warmup = args.warmup if args.warmup is not None else 2
        return cpp_worker(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    import itw_amd
    itw_amd.lib()                                   # fail loudly if the HIP library is missing
    cur_sha = itw_amd.source_sha256()
    if FAKE:
        dev = torch.device("cpu")
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU; there is no CPU path"
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)

    # a hang must end in an error the driver can read, not in its 1800 s limit: watchdog for the whole run, timeout on collectives
    faulthandler.dump_traceback_later(int(os.environ.get("ITW_BENCH_WATCHDOG_S", "1500")), exit=True)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")     # a failed / timed-out collective tears the process down
        os.environ.setdefault("NCCL_ASYNC_ERROR_HANDLING", "1")
        tmo = datetime.timedelta(seconds=int(os.environ.get("ITW_BENCH_DIST_TIMEOUT_S", "300")))
        if FAKE:
            dist.init_process_group("gloo", timeout=tmo)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        # CPU-side barrier for the time rank 0's child process owns the GPUs (an RCCL barrier would spin a kernel on every GPU)
        cpu_group = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=int(os.environ.get("ITW_BENCH_CPP_TIMEOUT_S", "600")) + 120))

    fmt, prof = WORKLOADS[args.workload]
    heavy = fmt in ("bc7", "bc6h")
    steps = args.steps if args.steps is not None else (10 if heavy else 50)
    warmup = args.warmup if args.warmup is not None else (2 if heavy else 5)

    from itw_amd import shard
    scaling = args.scaling or ("strong" if world > 1 else "weak")
    size = args.size or (16384 if (scaling == "strong" and world > 1) else 4096)

    def run_job(scaling, size, steps, warmup):
        """One timed job: every step = encode of this rank's band + all-gather of the output bands over xGMI (RCCL; in
        place).  The gather of step i runs on RCCL's stream while step i+1 encodes (two whole-image buffers,
        shard.BandPipeline); all gathers are waited for inside the timed region.  Returns (elapsed_s max over ranks, ...)."""
        # the partition: K interleaved sub-bands per rank where the geometry allows (strong scaling only; K = 1: one contiguous band)
        K = pieces_for(size, world, args.interleave) if (scaling == "strong" and world > 1) else 1
        geos = [plan(scaling, size, world, rank, fmt, k, K) for k in range(K)]
        geo = dict(geos[0])
        imgs = [make_band(fmt, scaling, size, g, rank) for g in geos]
        for g, im in zip(geos, imgs):
            assert im.shape[0] == g["rows"] and im.shape[1] == g["width"]
            assert g["band_bytes"] * world * K == g["total_bytes"], "bench bands must be equal (in-place all-gather): pick a size whose block rows divide by K * N"
        for k, g in enumerate(geos):
            assert g["band_off"] == (k * world + rank) * g["band_bytes"]
        encs = [make_encoder(itw_amd, fmt, prof, im, dev) for im in imgs]
        encode_list = [e[0] for e in encs]
        img, d_img = imgs[0], encs[0][1]
        geo["rows"] = sum(g["rows"] for g in geos)             # texel rows this rank encodes per step, all its pieces
        geo["pieces"] = K
        if os.environ.get("ITW_BENCH_CORRUPT_RANK") == str(rank):
            # test hook (tests/test_sharding_gloo.py): this rank damages its band after encoding it -- the verification must see it
            clean = encode_list[0]

            def corrupting(out, clean=clean):
                clean(out)
                out[:1] ^= 0xFF
            encode_list[0] = corrupting
        pipe = shard.InterleavedPipeline(geos[0]["band_bytes"], world, rank, dev, encode_list)
        for _ in range(warmup):
            pipe.step()
        pipe.drain()
        if dist is not None:
            dist.barrier()
        _sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.step()
        pipe.drain()
        _sync()
        if dist is not None:
            dist.barrier()
        _sync()
        elapsed = time.perf_counter() - t0
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed, geo, img, d_img, pipe

    elapsed, geo, img, d_img, pipe = run_job(scaling, size, steps, warmup)
    d_band = pipe.band[0]
    nblocks = (geo["width"] // 4) * (geo["rows"] // 4)          # blocks this rank encodes per step (all its pieces)

    # N > 1: is the gathered image the image?  (after the timed region; never inside it)
    verdict = None
    if world > 1:
        verdict = verify_gather(itw_amd, dist, pipe.full[(pipe.steps - 1) % pipe.depth], scaling, size, world, rank, fmt, prof, dev, geo.get("pieces", 1))

    # kernel-only duration on the launch stream (HIP events), for the roofline
    k_avg_ms, k_min_ms = time_kernel(itw_amd, fmt, prof, d_img, d_band, steps=max(3, min(steps, 20)), warmup=1)
    if geo.get("pieces", 1) > 1 and not FAKE:
        # K sub-bands per rank: the rank's kernel time per step is the sum over its pieces (equal sizes; piece 0 measured above, the
        # others re-derived from the same seeded surface here)
        for k in range(1, geo["pieces"]):
            gk = plan(scaling, size, world, rank, fmt, k, geo["pieces"])
            dk = torch.from_numpy(make_band(fmt, scaling, size, gk, rank)).to(dev)
            a, m = time_kernel(itw_amd, fmt, prof, dk, d_band, steps=max(3, min(steps, 20)), warmup=1)
            k_avg_ms += a; k_min_ms += m
            del dk
    elif geo.get("pieces", 1) > 1:
        k_avg_ms *= geo["pieces"]; k_min_ms *= geo["pieces"]
    per_rank_ms = [round(k_avg_ms, 4)]
    per_rank_gather_ms = None
    if dist is not None:
        # the gathers of one step on their own (nothing encoding): this rank's K in-place all-gathers, wall clock between synchronisations
        dist.barrier()
        _sync()
        tg = time.perf_counter()
        for k in range(pipe.pieces):
            dist.all_gather_into_tensor(pipe.groups[0][k], pipe.piece[0][k])
        _sync()
        gather_ms = (time.perf_counter() - tg) * 1e3
        t = torch.zeros(2 * world, dtype=torch.float64, device=dev)
        t[rank] = k_avg_ms
        t[world + rank] = gather_ms
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        per_rank_ms = [round(float(v), 4) for v in t[:world].tolist()]
        per_rank_gather_ms = [round(float(v), 4) for v in t[world:].tolist()]

    result = None
    if rank == 0:
        pixels = geo["width"] * geo["height"]                  # the whole job per step, all ranks
        alg = ALG_BYTES[fmt] * nblocks
        achieved = alg / (k_avg_ms * 1e-3) / 1e9
        result = {
            "metric": "Mpixels/s encode (BC1/BC3/BC7/BC6H) at 4k x 4k; bit-exact vs the oracle under the pinned ISPC sse/avx-target "
                      "arithmetic model (no ispc binary exists here; sensitivity of that model: profiles/arith_sensitivity.txt)",
            "value": round(pixels * steps / elapsed / 1e6, 2), "unit": "Mpixels/s",
            "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {fmt.upper()}" + (f" GetProfile_{prof}" if prof else "")
                       + f" on ONE synthetic {geo['width']}x{geo['height']} " + ("RGBA16F" if fmt == "bc6h" else "RGBA8")
                       + f" surface, {scaling}-sharded over {world} GPU(s) by block-row bands ({geo['rows']} texel rows per rank), "
                         "texels resident in HBM, device-pointer C ABI call",
                       "blocks_per_gpu": nblocks, "sharding": (f"{geo.get('pieces', 1)} interleaved block-row sub-band(s) per rank (sub-band j of K*N on rank j % N: "
                       "itwBandForPart; content-aware partition); one in-place all_gather per group of N sub-bands over RCCL, overlapped with the next encode")
                       if world > 1 else "single GPU",
                       "ranks_seen_by_rccl": (dist.get_world_size() if dist is not None else 1),
                       "rccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if (dist is not None and not FAKE) else None),
                       "bc7_mode_order": ("reference order (ITW_BC7_BOUND=0)" if os.environ.get("ITW_BC7_BOUND", "")[:1] == "0" else
                                          "per call, on the device: a pilot behind the first {0,2} scan estimates how many blocks still need modes 1/3 and picks the bounded order "
                                          "(modes 1/3 last, only where their exact lower bound allows) or the reference's for the rest of the call; two interleaved bands on two "
                                          "streams; same bytes either way; the time is content dependent: formats[*@baboon_tiled] is the natural-image end") if fmt == "bc7" else None,
                       "device": ("CONTROL-FLOW TEST ON CPU -- not a measurement" if FAKE else itw_amd.device_info()), "lib": itw_amd.version()},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "hbm_frac": round(achieved / HBM_PEAK_GBS, 5),
                         "traffic": pmc_traffic(args.workload, cur_sha) if (world == 1 and size == 4096) else None,   # PMC passes were taken at 4096^2
                         "kernel_ms_avg": round(k_avg_ms, 4), "kernel_ms_min": round(k_min_ms, 4),
                         "algorithmic_bytes_per_launch": alg,
                         "note": "BC7/BC6H are VALU-issue bound (no MFMA-shaped work); the HBM fraction is reported "
                                 "because the contract asks for it; `valu` is the roofline that binds (DESIGN.md 3)"},
        }
        if verdict is not None:
            result.update(verdict)
            result["per_rank_kernel_ms"] = per_rank_ms
            # N > 1: what the first hardware run is read by -- per rank, and how unequal the ranks are
            result["config"]["per_rank_encode_ms"] = per_rank_ms
            result["config"]["per_rank_gather_ms"] = per_rank_gather_ms
            result["config"]["max_over_mean"] = round(max(per_rank_ms) / (sum(per_rank_ms) / len(per_rank_ms)), 4) if min(per_rank_ms) > 0 else None
            result["config"]["sub_bands_per_rank"] = geo.get("pieces", 1)
            # the same job on ONE GPU, from the latest committed single-GPU run of this geometry: what a strong-scaling ratio of
            # this line should be taken against (the default N = 1 line is BASELINE configs[2], a 4096^2 surface -- another job)
            try:
                names = sorted(n for n in os.listdir(os.path.join(ROOT, "profiles")) if n.endswith(f"_bench_{size}_strong_n1.json"))
                with open(os.path.join(ROOT, "profiles", names[-1])) as f:
                    one = json.load(f)
                result["same_job_on_one_gpu"] = {"value": one["value"], "unit": one["unit"], "ms_per_step": one["ms_per_step"],
                                                 "source": "profiles/" + names[-1]}
            except (OSError, ValueError, KeyError, IndexError):
                result["same_job_on_one_gpu"] = None
        # Counter values below come from COMMITTED profiling passes, not from this run: each pass carries the SHA-256 of the kernel
        # sources it was taken on (tools/profile_gpu.sh) and is quoted only when that equals the tree this runs from
        rk = rocprof_kernels(fmt)
        insts, per_kernel, valu_sha = pmc_valu(args.workload)
        at_profile_size = world == 1 and size == 4096 and scaling == "weak"
        if at_profile_size:
            stamps = {"pmc_traffic": (_latest_profile("pmc_traffic.json")[1] or {}).get("_source_sha256"), "valu": valu_sha,
                      "rocprof_kernels": rk.get("source_sha256") if rk else None}
            result["roofline"]["profile_matches_source"] = {k: (v == cur_sha) for k, v in stamps.items()}
            result["roofline"]["source_sha256"] = cur_sha
        if rk and at_profile_size and rk.get("source_sha256") == cur_sha:
            # the call is several kernels for BC7; `achieved` uses the whole call.  For the slow / alpha_slow profiles the
            # kernels listed are exactly the call's (the ranked variants <.., 1|2, ..> belong to the faster presets).
            result["roofline"]["rocprof_kernels"] = rk
        iso = isolated_kernels(args.workload, cur_sha) if at_profile_size else None
        if iso:
            result["roofline"].setdefault("rocprof_kernels", {})["isolated"] = iso
        if insts and at_profile_size and valu_sha == cur_sha:
            result["roofline"]["valu"] = valu_block(insts, k_avg_ms)
            result["roofline"]["valu"]["note"] = (
                "SQ_INSTS_VALU of one call (committed rocprofv3 pass) x 64 lanes / live kernel time; peak = 256 CU x 4 SIMD "
                "x 32 lanes x 2.4 GHz, reached only by the 2-cycle VOP2 forms (v_mul/add_f32, v_add_u32, logic, shifts right); "
                "cvt / cmp / fma / packed / dot forms issue at half that rate (tools/ubench): `issue` prices every form")
            issue = issue_block(per_kernel, insts, k_avg_ms, cur_sha)
            if issue:
                result["roofline"]["issue"] = issue
            # the binding roofline as scalars (the driver's record keeps scalar roofline fields only): BC7 / BC6H are VALU-issue bound
            result["roofline"]["valu_frac"] = result["roofline"]["valu"]["frac"]
            result["roofline"]["issue_frac"] = issue["frac"] if issue else None
            if fmt in ("bc7", "bc6h"):
                result["roofline"]["bound"] = "valu-issue"
                result["roofline"]["frac_is"] = ("hbm_frac (the contract's field: algorithmic bytes / kernel time / 8 TB/s); the fractions of the roofline "
                                                 "that binds are valu_frac (lane-ops / 78.6 T) and issue_frac (issue cycles needed / available)")
        else:
            insts = None

    if rank == 0:
        per_block, method = op_counts(args.workload)
        if per_block:
            # algorithmic roofline: the reference's own fp32 work (measured on the oracle) over this GPU's time
            blocks_all = (geo["width"] // 4) * (geo["height"] // 4)
            arith, full = per_block.get("fp32_arith_ops", 0.0), per_block.get("fp32_arith_cmp_cvt_ops", 0.0)
            t = elapsed / steps
            ref_rate = full * blocks_all / t / 1e12
            va = {"fp32_arith_ops_per_block": arith, "fp32_arith_cmp_cvt_ops_per_block": full,
                  "reference_T_ops_s": round(ref_rate, 2), "peak_T_flops_s": VALU_PEAK_FMA_TFLOPS * world,
                  "reference_equivalent_over_fma_peak": round(ref_rate / (VALU_PEAK_FMA_TFLOPS * world), 4),
                  "note": "NOT a roofline of this implementation, and NOT bounded by 1: since round 4's bounded BC7 mode order the kernels "
                          "leave out, exactly, the parts of the reference's work that cannot change the block (DESIGN.md 3.2), so the "
                          "reference-EQUIVALENT rate can exceed the chip's peak.  It is the reference algorithm's scalar fp32 operations per block (mul/add/"
                          "div/sqrt + compares + conversions, counted by single-stepping the CPU oracle, profiles/op_counts.json) x blocks "
                          "/ measured step time, against the chip's fp32 vector peak WITH FMA / packed credit (157.3 TFLOP/s per GPU: "
                          "one lane-op = up to 2 flops).  The kernels retire several reference ops per executed lane-op (a v_dot4 "
                          "= 7, a v_pk_fma_f32 = 4): `reference_ops_per_executed_lane_op` is that ratio; `roofline.valu` is the "
                          "roofline that binds."}
            if insts and world == 1 and size == 4096 and scaling == "weak":
                va["reference_ops_per_executed_lane_op"] = round(full * blocks_all / (insts * 64), 3)
            result["roofline"]["valu_algorithmic"] = va

    if world > 1 and scaling == "strong":
        # side figure: weak scaling, one 4096^2 band per rank (what round 1 reported); a few steps only
        del d_img, pipe
        if not FAKE:
            torch.cuda.empty_cache()
        w_steps = max(3, min(steps, 5))
        w_elapsed, w_geo, _, _, w_pipe = run_job("weak", 4096, w_steps, 1)
        if rank == 0:
            result["weak_side"] = {"value": round(w_geo["width"] * w_geo["height"] * w_steps / w_elapsed / 1e6, 2), "unit": "Mpixels/s",
                                   "ms_per_step": round(w_elapsed / w_steps * 1e3, 4), "steps": w_steps,
                                   "workload": f"{args.workload} on a 4096 x {4096 * world} surface, one 4096^2 band per rank"}
        del w_pipe

    # N > 1: the same job through the C++ host path (one process, itwCompressImageMultiGPUEx) -- the headline when it succeeds
    want_cpp = world > 1 and scaling == "strong" and args.host in ("auto", "cpp") and (not FAKE or os.environ.get("ITW_BENCH_FAKE_CPP"))
    if want_cpp:
        if not FAKE:
            torch.cuda.empty_cache()
            _sync()
        dist.barrier(group=cpu_group)
        cpp = run_cpp_child(args, world, size, steps, warmup) if rank == 0 else None
        dist.barrier(group=cpu_group)                         # ranks 1..N-1 wait on the CPU while the child owns the GPUs
        if rank == 0:
            python_side = {k: result.get(k) for k in ("value", "unit", "ms_per_step", "gather_verified", "mismatching_bytes", "band_checks",
                                                      "per_rank_kernel_ms", "how")}
            python_side["host"] = "python: one process per GPU, torch.distributed all_gather_into_tensor (RCCL), gather of step i under encode i+1"
            python_side["ranks_seen_by_rccl"] = result["config"]["ranks_seen_by_rccl"]
            result["python_side"] = python_side
            if "error" not in cpp and cpp.get("gather_verified"):
                for k in ("value", "ms_per_step", "gather_verified", "mismatching_bytes", "band_checks", "how"):
                    result[k] = cpp.get(k)
                result["steps"], result["warmup"] = cpp["steps"], cpp["warmup"]
                result["config"]["host"] = cpp["host"]
                result["config"]["sharding"] = ("block-row bands, one per rank (itwBandForPart), band r resident on GPU r; gather of the block stream "
                                                "to GPU 0 by RCCL ncclSend / grouped ncclRecv on a second stream, a rank's first half-band under its second half's encode")
                result["config"]["ranks_seen_by_rccl"] = cpp["ranks_seen_by_rccl"]
                result["config"]["transport"] = cpp["transport"]
                pr = (cpp.get("stats_last_call") or {}).get("per_rank") or []
                if pr:                                         # the headline job's own account (device-side HIP events per rank)
                    enc = [r["encode_ms"] for r in pr]
                    result["config"]["per_rank_encode_ms"] = enc
                    result["config"]["per_rank_gather_ms"] = [r["gather_ms"] for r in pr]
                    result["config"]["max_over_mean"] = round(max(enc) / (sum(enc) / len(enc)), 4) if min(enc) > 0 else None
                    result["config"]["sub_bands_per_rank"] = cpp.get("sub_bands_per_rank")
                result["timing"] = ("C++ job: K synchronous itwCompressImageMultiGPUEx calls (every rank stream drained before a call returns) between two "
                                    "all-device synchronisations, wall clock of the one process that drives all N GPUs; python_side: barrier + "
                                    "synchronize on both sides, MAX over ranks")
                result["cpp_host"] = {k: cpp.get(k) for k in ("transport", "transport_note", "ranks_seen_by_rccl", "peer_links", "devices",
                                                              "first_call_wall_ms", "stats_last_call", "scatter_from_gpu0")}
            else:
                cpp["note"] = "the C++ job did not produce a verified result: the headline is the torch.distributed job (python_side)"
                result["cpp_host"] = cpp
                result["config"]["host"] = python_side["host"]

    size_side = 4096
    if rank == 0 and world == 1 and not args.no_formats and size == 4096:
        size, nblocks = size_side, (size_side // 4) ** 2
        side = {}
        # the presets IntelPlugin.cpp:832-843 selects (veryfast / basic / alpha_veryfast / alpha_basic, BC6H fast / slow) ride along with the slow ones
        for wl in ("bc1", "bc3", "bc4", "bc5", "bc7_veryfast", "bc7_basic", "bc7_alpha_veryfast", "bc7_alpha_basic", "bc7_slow", "bc7_alpha_slow", "bc6h_fast", "bc6h_slow"):
            if wl == args.workload:
                continue
            f2, p2 = WORKLOADS[wl]
            try:
                im2 = make_surface(f2, size, 0)
                d2 = torch.from_numpy(im2).to(dev)
                o2 = torch.empty(nblocks * itw_amd.BYTES_PER_BLOCK[f2], dtype=torch.uint8, device=dev)
                light = f2 not in ("bc7", "bc6h")          # microsecond kernels: enough launches for clocks and TLBs to settle
                avg, mn = time_kernel(itw_amd, f2, p2, d2, o2, steps=200 if light else 3, warmup=20 if light else 1, back_to_back=light)
                gbs = ALG_BYTES[f2] * nblocks / (avg * 1e-3) / 1e9
                side[wl] = {"Mpixels/s": round(size * size / (avg * 1e-3) / 1e6, 1), "kernel_ms_avg": round(avg, 4),
                            "hbm_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 5),
                            "traffic": pmc_traffic(wl, cur_sha)}
                i2, pk2, sha2 = pmc_valu(wl)
                if i2 and sha2 == cur_sha:                     # the binding roofline of every format (SURVEY 8d asks for both)
                    side[wl]["valu"] = valu_block(i2, avg)
                    iss = issue_block(pk2, i2, avg, cur_sha)
                    if iss:
                        side[wl]["issue"] = {k: iss[k] for k in ("cycles_needed", "cycles_available", "frac", "model_matches_source")}
                if light:
                    side[wl]["timing"] = "one HIP event pair around 200 back-to-back launches"
                if wl == "bc7_alpha_slow":
                    # the RGBA profiles' cost depends on the alpha channel (alpha-group-first order): the synthetic
                    # surface's alpha is translucent everywhere (best case); the same RGB with opaque and per-block mixed alpha
                    side[wl]["content"] = "alpha translucent in every block (survey input I3): RGB modes are skipped almost everywhere"
                    from itw_amd import surfaces
                    for kind in ("opaque", "mixed"):
                        d3 = torch.from_numpy(surfaces.ldr_alpha_variant(im2, kind)).to(dev)
                        a3, _ = time_kernel(itw_amd, f2, p2, d3, o2, steps=3, warmup=1)
                        side[f"{wl}@{kind}"] = {"Mpixels/s": round(size * size / (a3 * 1e-3) / 1e6, 1), "kernel_ms_avg": round(a3, 4),
                                                "content": "alpha = 255 everywhere" if kind == "opaque" else "per 4x4 block: opaque or translucent, 50/50"}
                        del d3
                del d2, o2
            except Exception as e:  # a format whose kernel is not built yet aborts in C; anything else lands here
                side[wl] = {"error": repr(e)}
        if size == 4096:
            # SURVEY 8(d) input I2: the reference's colors-16M.png (regenerated by formula), same 4096^2 block count
            try:
                from itw_amd import surfaces
                d2 = torch.from_numpy(surfaces.colors_16m()).to(dev)
                o2 = torch.empty(nblocks * 16, dtype=torch.uint8, device=dev)
                for wl in ("bc1", "bc7_slow"):
                    f2, p2 = WORKLOADS[wl]
                    avg, mn = time_kernel(itw_amd, f2, p2, d2, o2, steps=3 if f2 == "bc7" else 200, warmup=1 if f2 == "bc7" else 20, back_to_back=f2 != "bc7")
                    side[wl + "@colors16m"] = {"Mpixels/s": round(size * size / (avg * 1e-3) / 1e6, 1), "kernel_ms_avg": round(avg, 4)}
                del d2, o2
            except Exception as e:
                side["@colors16m"] = {"error": repr(e)}
            # BC7 is content dependent since round 4's bounded mode order (modes 1/3 only where their exact lower bound is below the other
            # modes' result: csrc/bc7.hip): the headline surface is noisy and skips most of them; a natural image does not.  I1 of
            # SURVEY 8(d) (the reference's baboon.png, tests/golden/inputs.npz) tiled to the same block count shows the other end.
            try:
                z = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
                reps = -(-size // z["baboon"].shape[0])
                nat = np.ascontiguousarray(np.tile(z["baboon"], (reps, reps, 1))[:size, :size])
                d2 = torch.from_numpy(nat).to(dev)
                o2 = torch.empty(nblocks * 16, dtype=torch.uint8, device=dev)
                for wl in ("bc7_slow", "bc7_alpha_slow"):
                    f2, p2 = WORKLOADS[wl]
                    avg, mn = time_kernel(itw_amd, f2, p2, d2, o2, steps=3, warmup=1)
                    side[wl + "@baboon_tiled"] = {"Mpixels/s": round(size * size / (avg * 1e-3) / 1e6, 1), "kernel_ms_avg": round(avg, 4),
                                                   "content": "the reference's baboon.png (opaque) tiled to the surface: modes 1/3 win 60 % of its blocks, nothing is skipped"}
                del d2, o2
                # what the reference's callers pass: HOST pointers (pageable memory), one synchronous call incl. H2D + D2H (never bench.py's
                # value).  The runs of such a call are overlapped deep bands, or -- where the pilot's estimate says nearly every block needs
                # modes 1/3 -- the wide shape run after run (csrc/abi.hip); best of 5 calls after a warm-up call
                for tag, himg in (("synthetic", make_surface("bc7", size, 0)), ("baboon_tiled", nat)):
                    itw_amd.compress_numpy("bc7", himg, "slow")
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter()
                        itw_amd.compress_numpy("bc7", himg, "slow")
                        ts.append(time.perf_counter() - t0)
                    side["bc7_slow@host_pointers_" + tag] = {"call_ms_best_of_5": round(min(ts) * 1e3, 3), "Mpixels/s": round(size * size / min(ts) / 1e6, 1),
                                                             "what": "one synchronous CompressBlocksBC7 call with pageable host pointers: upload + kernels + download"}
            except Exception as e:
                side["@baboon_tiled"] = {"error": repr(e)}
            # the plugin's own calling pattern -- 64 slices with SetProgress between them -- through the pipelined slice loop
            side["sliced@64"] = sliced_side(itw_amd, size, make_surface)
            # BC1 / BC3 are the HBM-side kernels: the same kernel on the 16384^2 surface of configs[4], where the launch ramp and
            # tail (about 6 us) stop mattering -- the steady-state fraction of the HBM roofline
            try:
                if args.no_16k:
                    raise StopIteration
                big = 16384
                d2 = torch.from_numpy(make_surface("bc1", 4096, 0)).to(dev).repeat(big // 4096, big // 4096, 1).contiguous()   # I5 = I3 tiled
                o2 = torch.empty((big // 4) ** 2 * 16, dtype=torch.uint8, device=dev)
                for wl in ("bc1", "bc3"):
                    iso, _ = time_kernel(itw_amd, wl, None, d2, o2, steps=40, warmup=10)
                    avg, _ = time_kernel(itw_amd, wl, None, d2, o2, steps=40, warmup=0, back_to_back=True)
                    gbs = ALG_BYTES[wl] * (big // 4) ** 2 / (avg * 1e-3) / 1e9
                    side[wl + "@16384"] = {"Mpixels/s": round(big * big / (avg * 1e-3) / 1e6, 1), "kernel_ms_avg": round(avg, 4),
                                           "hbm_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 5),
                                           "kernel_ms_isolated": round(iso, 4),
                                           "timing": "kernel_ms_avg: one HIP event pair around 40 back-to-back launches (as the 4096^2 "
                                                     "figures); kernel_ms_isolated: an event pair per launch (each pays its own ramp and drain)"}
                del d2, o2
                torch.cuda.empty_cache()
            except StopIteration:
                pass
            except Exception as e:
                side["@16384"] = {"error": repr(e)}
            # The C++ multi-GPU entry on this one device: the 16384^2 `slow` surface of configs[4] through 8
            # virtual ranks (8 host threads x 2 half-bands on 8 stream pairs of the same GPU, encoded in place) against ONE call --
            # what the dispatch layer costs; same bytes required
            try:
                if args.no_16k:
                    raise StopIteration
                big = 16384
                d2 = torch.from_numpy(make_surface("bc7", 4096, 0)).to(dev).repeat(big // 4096, big // 4096, 1).contiguous()
                o2 = torch.empty((big // 4) ** 2 * 16, dtype=torch.uint8, device=dev)
                o3 = torch.zeros_like(o2)
                one_avg, one_min = time_kernel(itw_amd, "bc7", "slow", d2, o2, steps=3, warmup=1)
                st = itw_amd.MultiGpuStats()
                itw_amd.compress_image_multigpu("bc7", d2, "slow", ranks=8, out=o3, stats=st)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    itw_amd.compress_image_multigpu("bc7", d2, "slow", ranks=8, out=o3, stats=st)
                torch.cuda.synchronize()
                multi_ms = (time.perf_counter() - t0) / 3 * 1e3
                sd = st.as_dict()
                side["multigpu_cpp@8virtual_16384"] = {
                    "what": "bc7 slow, one 16384^2 surface: itwCompressImageMultiGPUEx with 8 ranks on this one device vs one CompressBlocksBC7 call",
                    "single_call_ms": round(one_avg, 3), "multigpu_8virtual_ms": round(multi_ms, 3), "overhead_frac": round(multi_ms / one_avg - 1.0, 4),
                    "Mpixels/s": round(big * big / (multi_ms * 1e-3) / 1e6, 1), "identical_bytes": bool(torch.equal(o2, o3)),
                    "transport": sd["transport"], "transport_note": sd["transport_note"], "posted_ms": sd["posted_ms"], "wall_ms_last_call": sd["wall_ms"],
                    "per_rank_encode_ms": [r["encode_ms"] for r in sd["per_rank"]],
                    "timing": "single call: HIP events on its stream, 3 calls; multi: wall clock of 3 synchronous calls (includes waking 8 host threads)"}
                del d2, o2, o3
                torch.cuda.empty_cache()
            except StopIteration:
                pass
            except Exception as e:
                side["multigpu_cpp@8virtual_16384"] = {"error": repr(e)}
            # the partition on MIXED content.  A 16384^2 mosaic -- top half photographs (baboon.png and monkey.png
            # tiled: nearly every block still visits modes 1/3), bottom half the synthetic surface (30 % do) -- cut (a) into 8 contiguous
            # bands, the reference's rule, and (b) into 32 sub-bands dealt round-robin to 8 ranks (K = 4, the library's default).  Every
            # (sub-)band is encoded ALONE on this GPU (HIP events): a rank's load is the sum over its pieces, independent of how the
            # ranks would share a device.  Then the same image through itwCompressImageMultiGPUEx with both partitions: same bytes.
            try:
                if args.no_16k:
                    raise StopIteration
                big = 16384
                gi = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))

                def tile4096(a):
                    a = a[:a.shape[0] // 4 * 4, :a.shape[1] // 4 * 4]
                    return np.ascontiguousarray(np.tile(a, (-(-4096 // a.shape[0]), -(-4096 // a.shape[1]), 1))[:4096, :4096])
                nat = [torch.from_numpy(tile4096(gi[k])).to(dev) for k in ("baboon", "monkey")]
                syn = torch.from_numpy(make_surface("bc7", 4096, 0)).to(dev)
                rows = [torch.cat([nat[(i + j) % 2] for j in range(4)], dim=1) for i in range(2)] + [torch.cat([syn] * 4, dim=1)] * 2
                d2 = torch.cat(rows, dim=0).contiguous()
                d2[..., 3] = 255
                del nat, syn, rows
                o_ref = torch.empty((big // 4) ** 2 * 16, dtype=torch.uint8, device=dev)
                itw_amd.compress("bc7", d2, "slow", out=o_ref)
                torch.cuda.synchronize()

                def loads(parts, ranks):
                    ms = []
                    for j in range(parts):
                        y0, nrows, off = itw_amd.band_for_part(big, big, "bc7", j, parts)
                        a, _ = time_kernel(itw_amd, "bc7", "slow", d2[y0:y0 + nrows], o_ref[off:off + (nrows // 4) * (big // 4) * 16], steps=2, warmup=1)
                        ms.append(a)
                    per_rank = [sum(ms[j] for j in range(parts) if j % ranks == r) for r in range(ranks)]
                    return ms, per_rank
                ms8, load8 = loads(8, 8)
                ms32, load32 = loads(32, 8)
                o3 = torch.zeros_like(o_ref)
                wall = {}
                same = True
                for K in (1, 4):
                    itw_amd.compress_image_multigpu("bc7", d2, "slow", ranks=8, out=o3, interleave=K)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(2):
                        itw_amd.compress_image_multigpu("bc7", d2, "slow", ranks=8, out=o3)
                    torch.cuda.synchronize()
                    wall[K] = (time.perf_counter() - t0) / 2 * 1e3
                    same = same and bool(torch.equal(o3, o_ref))
                    o3.zero_()
                itw_amd.lib().itwMultiGpuSetInterleave(4)
                side["multigpu_cpp@8virtual_16384_mixed"] = {
                    "what": "bc7 slow on a 16384^2 mosaic (top half baboon.png / monkey.png tiled, bottom half the synthetic surface): per-rank load of 8 ranks under the "
                            "reference's contiguous bands vs 4 interleaved sub-bands per rank (itwMultiGpuSetInterleave; the default), each (sub-)band encoded alone on this GPU",
                    "contiguous_band_ms": [round(v, 3) for v in ms8], "contiguous_max_over_mean": round(max(load8) / (sum(load8) / 8), 4),
                    "interleaved_rank_ms": [round(v, 3) for v in load32], "interleaved_max_over_mean": round(max(load32) / (sum(load32) / 8), 4),
                    "sum_of_bands_ms": {"contiguous": round(sum(ms8), 3), "interleaved": round(sum(ms32), 3)},
                    "multigpu_8virtual_wall_ms": {"contiguous": round(wall[1], 3), "interleaved": round(wall[4], 3)},
                    "identical_bytes": same,
                    "reading": "on N devices the slowest rank sets the time: the ideal N-way speed-up is divided by max/mean"}
                del d2, o_ref, o3
                torch.cuda.empty_cache()
            except StopIteration:
                pass
            except Exception as e:
                side["multigpu_cpp@8virtual_16384_mixed"] = {"error": repr(e)}
            # SURVEY 8(d) input I4 / BASELINE configs[3]: the reference's monkey-32bit.hdr (RGBE -> RGBA16F, committed as a
            # fixture: tests/golden/inputs.npz) tiled 19 x 19 and cropped to 4096^2
            try:
                from itw_amd import surfaces
                mk = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))["monkey_hdr"]
                d2 = torch.from_numpy(surfaces.tile_to(mk, size, size)).to(dev)
                o2 = torch.empty(nblocks * 16, dtype=torch.uint8, device=dev)
                for wl in ("bc6h_fast", "bc6h_slow"):
                    f2, p2 = WORKLOADS[wl]
                    avg, mn = time_kernel(itw_amd, f2, p2, d2, o2, steps=3, warmup=1)
                    gbs = ALG_BYTES[f2] * nblocks / (avg * 1e-3) / 1e9
                    side[wl + "@monkey_hdr_tiled"] = {"Mpixels/s": round(size * size / (avg * 1e-3) / 1e6, 1), "kernel_ms_avg": round(avg, 4),
                                                      "hbm_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 5)}
                del d2, o2
            except Exception as e:
                side["@monkey_hdr_tiled"] = {"error": repr(e)}
        if not args.no_cpu:
            # the CPU path beside every format: scalar C oracle, 1 thread and all usable threads, ~2 s each
            for wl in ("bc1", "bc3", "bc6h_slow"):
                f2, p2 = WORKLOADS[wl]
                if wl in side and "error" not in side[wl]:
                    try:
                        side[wl]["cpu_baseline"] = cpu_baseline_pair(f2, p2, make_surface(f2, 1024 if wl == "bc6h_slow" else 4096, 0))
                    except Exception as e:
                        side[wl]["cpu_baseline"] = {"error": repr(e)}
        result["formats"] = side

    if rank == 0 and world == 1 and not args.no_cpu:
        result["cpu_baseline"] = cpu_baseline(fmt, prof, img)
    elif rank == 0:
        result["cpu_baseline"] = None

    if rank == 0:
        result["abi_calls"] = ABI_CALLS[0]      # encode calls of this process (headline legs; tools/summarize_profiles.py normalises profiler counters by it)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    faulthandler.cancel_dump_traceback_later()


if __name__ == "__main__":
    main()
