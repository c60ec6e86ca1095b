#!/bin/bash
# Round 4, first GPU pass: the new C++ multi-GPU entry (stats, resident bands, watchdog), the BC4/BC5 run table, the mode 4/5 scalar
# skip (A/B against gpurun_variants/lib_noskip.so), the bench line with the 8-virtual-rank figure.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_multigpu_cpp.py tests/test_gpu_parity_bc4_bc5.py tests/test_reference_codecs.py tests/test_gpu_bench_contract.py -m gpu -q -x > $O/pytest_new.log 2>&1; tail -3 $O/pytest_new.log
timeout 900 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py -m gpu -q -x > $O/pytest_bc7.log 2>&1; tail -3 $O/pytest_bc7.log
timeout 600 python tools/parity_campaign.py 8 oracle bc4,bc5 > $O/parity_campaign_bc45_8Mpix.txt 2>&1; tail -3 $O/parity_campaign_bc45_8Mpix.txt
bash tools/gpu_variants.sh > $O/variants.txt 2>&1; cat $O/variants.txt
timeout 600 python tools/profile_table.py 2>&1 | grep -v amdgpu > $O/preset_table.txt; grep -E "^bc[1345] " $O/preset_table.txt
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04a/bench_default.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"])
f = j["formats"]
for k in ("bc1", "bc3", "bc4", "bc5", "multigpu_cpp@8virtual_16384"):
    print(k, json.dumps(f.get(k))[:600])
PY
