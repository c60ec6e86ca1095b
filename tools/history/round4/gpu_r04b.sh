#!/bin/bash
# Round 4, second GPU pass: BC4/BC5 with the v_perm index lookup, the wide shape's split width (WIDE_FILL_DIVISOR 2 vs 1),
# the bench line with stamped counters.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_bc4_bc5.py tests/test_reference_codecs.py -m gpu -q -x > $O/pytest_bc45.log 2>&1; tail -2 $O/pytest_bc45.log
timeout 600 python tools/parity_campaign.py 8 oracle bc4,bc5 > $O/parity_campaign_bc45_8Mpix.txt 2>&1; tail -1 $O/parity_campaign_bc45_8Mpix.txt
timeout 300 python tools/profile_table.py 2>&1 | grep -E "^bc[1345] " | tee $O/preset_table_bc1345.txt
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
for v in orig fill1; do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp gpurun_variants/lib_$v.so $L; fi
  echo "== $v"
  timeout 600 python tools/bc7_path_probe.py slow,alpha_slow,basic 2>&1 | grep -v amdgpu
done > $O/wide_fill_ab.txt 2>&1
cp /tmp/orig.so $L
cat $O/wide_fill_ab.txt
timeout 600 python -m pytest tests/test_gpu_bc7_paths.py -m gpu -q -x 2>&1 | tail -2
timeout 900 python bench.py --no-cpu > $O/bench_nocpu.json 2> $O/bench_nocpu.err; tail -c 1500 $O/bench_nocpu.err; python - <<'PY'
import json
j = json.loads([l for l in open("gpurun_out/r04b/bench_nocpu.json") if l.startswith("{")][-1])
print("value", j["value"], "ms", j["ms_per_step"], json.dumps(j["roofline"])[:900])
f = j["formats"]
for k in ("bc4", "bc5"):
    print(k, json.dumps(f.get(k))[:400])
PY
