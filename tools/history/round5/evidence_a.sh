#!/bin/bash
# Round 5 evidence, part A (one gpurun call): tests, parity campaigns, timing tables, the bench lines.  -> gpurun_out/evidence_r05/
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/evidence_r05; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 900 python tools/parity_campaign.py 2 > $OUT/parity_campaign_2Mpix_wide.txt 2>&1; tail -1 $OUT/parity_campaign_2Mpix_wide.txt
timeout 1500 python tools/parity_campaign.py 16 oracle bc7 slow,alpha_slow,basic > $OUT/parity_campaign_16Mpix_bc7_fused_pilot_bands.txt 2>&1; tail -1 $OUT/parity_campaign_16Mpix_bc7_fused_pilot_bands.txt
timeout 1500 python tools/parity_campaign.py 4 ref bc7,bc1,bc3,bc6h > $OUT/parity_campaign_4Mpix_vs_reference_kernel.txt 2>&1; tail -1 $OUT/parity_campaign_4Mpix_vs_reference_kernel.txt
timeout 900 python tools/gpu_settings_fuzz.py 400 5 > $OUT/gpu_settings_fuzz_400.txt 2>&1; tail -1 $OUT/gpu_settings_fuzz_400.txt
timeout 600 python tools/profile_table.py 2>&1 | grep -v amdgpu > $OUT/preset_table.txt
timeout 600 python tools/bc7_path_probe.py slow,basic,alpha_basic,veryfast,alpha_slow 2>&1 | grep -v amdgpu > $OUT/bc7_path_probe.txt
timeout 600 python tools/host_path_timing.py 2>&1 | grep -v amdgpu > $OUT/host_pointer_path.txt
timeout 900 python tools/ref_caller_timing.py 4096 8,64 > $OUT/reference_caller_timing.jsonl 2>&1
ORDER_PROFILES=slow,alpha_slow timeout 600 python tools/round5/order_timing.py I3 I2 baboon test_a mixed monkey landscape 2>&1 | grep -v amdgpu > $OUT/bc7_order_policy_by_content.txt
ITW_BC7_PILOT_DEBUG=1 ORDER_HOST=0 timeout 300 python tools/round5/order_timing.py I3 I2 baboon test_a mixed monkey landscape 2>&1 | grep "^bc7 pilot\|^==" | uniq -c > $OUT/bc7_pilot_verdicts_by_content.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 900 python bench.py --size 16384 --scaling strong --steps 5 --warmup 1 --no-formats --no-cpu > $OUT/bench_16384_strong_n1.json 2>> $OUT/bench_default.err
cd /tmp
for c in I3 baboon; do
  ORDER_HOST=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/tr_$c -o t -- python $GRAFT_REPO_ROOT/tools/round5/order_timing.py $c > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$OUT/tr_$c -name '*kernel_trace*.csv' | head -1)
  echo "== $c (default policy), one call" >> $GRAFT_REPO_ROOT/$OUT/bc7_call_timeline.txt
  python $GRAFT_REPO_ROOT/tools/round5/trace_timeline.py $f bc7_pilot_estimate >> $GRAFT_REPO_ROOT/$OUT/bc7_call_timeline.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/$OUT/tr_$c
done
cd $GRAFT_REPO_ROOT
ls -la $OUT
