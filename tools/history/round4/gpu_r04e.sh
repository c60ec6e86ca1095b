#!/bin/bash
# Round 4: the C++ multi-GPU bench worker end to end on the one device (8 ranks sharing it), incl. the scatter variant
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
ITW_BENCH_CPP_SHARE_DEVICES=1 timeout 900 python bench.py --cpp-worker --gpus 8 --size 16384 --steps 3 --warmup 1 > $O/cpp_worker_8virtual.json 2> $O/cpp_worker.err; tail -c 600 $O/cpp_worker.err
python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04e/cpp_worker_8virtual.json") if l.startswith("{")][-1])
for k in ("value", "ms_per_step", "gather_verified", "mismatching_bytes", "band_checks", "transport", "transport_note", "first_call_wall_ms"):
    print(k, j.get(k))
print("scatter", {k: j["scatter_from_gpu0"][k] for k in ("ms_per_step", "value", "identical_to_resident_bands_result")})
print("rank0", j["stats_last_call"]["per_rank"][0])
PY
