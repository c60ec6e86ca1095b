# round 5, batch h: the whole GPU suite (new multi-GPU partition tests), then the default bench line
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 | tee $O/pytest_gpu.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05h/bench_default.json"))
print({k: j[k] for k in ("value", "ms_per_step")}, j["roofline"]["kernel_ms_avg"], j.get("cpu_baseline", {}).get("value"))
for k, v in j.get("formats", {}).items():
    print(k, {kk: vv for kk, vv in v.items() if kk in ("Mpixels/s", "kernel_ms_avg", "hbm_frac", "error", "contiguous_max_over_mean", "interleaved_max_over_mean", "identical_bytes", "multigpu_8virtual_wall_ms", "sum_of_bands_ms", "single_call_ms", "multigpu_8virtual_ms", "overhead_frac")}, (v.get("cpu_baseline") or {}).get("threads_1"), (v.get("cpu_baseline") or {}).get("threads_all"))
PY
