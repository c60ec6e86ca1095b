#!/bin/bash
# ONE script that regenerates a round's committed evidence on the GPU box:   gpurun -- 'bash tools/evidence.sh <tag> [parts]'
#   parts (default "tests tables bench profile matrix"):
#     tests    pytest -m gpu (whole suite), parity campaigns, settings fuzz                    -> gpurun_out/evidence_<tag>/{pytest_gpu.log, parity_*, gpu_settings_fuzz_*}
#     tables   preset table, variant table, launch-shape probe, host-pointer calls, the plugin's slice loop (pipeline / literal / one call),
#              the reference's own win32Threads.cpp driving the library, mode-order policy by content, one call's kernel timeline
#     bench    python bench.py (default flags) and the 16384^2 geometry on one GPU
#     profile  tools/profile_gpu.sh <tag>: rocprofv3 kernel stats, PMC traffic, SQ counters (every plugin preset included)
#     long     the long campaigns: 16 Mpix of every format and preset, 64 Mpix per BC7 slow profile (4 M blocks per call), 1 200 + 1 200 random
#              settings structs, and a second seed of the 4 Mpix campaign against the reference's own kernel source
#     matrix   tools/gpu_env_matrix.sh: the BC7 / BC6H / dispatch / host-pointer suites under every environment switch
# Afterwards, in the build container:  python tools/summarize_profiles.py <tag>  &&  python tools/collect_evidence.py <tag>
# copy the summaries into profiles/<tag>_* (counter files into profiles/<tag>_counters/).  Lab scripts of earlier rounds: tools/history/.
set -u
TAG=${1:-r06}; shift || true
PARTS=${@:-tests tables bench profile matrix}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/evidence_$TAG; mkdir -p $OUT
export TMPDIR=/tmp
nog() { grep -v amdgpu; }
for part in $PARTS; do
case $part in
tests)
  timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
  timeout 900 python tools/parity_campaign.py 2 > $OUT/parity_campaign_2Mpix_wide.txt 2>&1; tail -1 $OUT/parity_campaign_2Mpix_wide.txt
  timeout 1500 python tools/parity_campaign.py 16 oracle bc7 slow,alpha_slow,basic,alpha_basic > $OUT/parity_campaign_16Mpix_bc7_fused.txt 2>&1; tail -1 $OUT/parity_campaign_16Mpix_bc7_fused.txt
  timeout 1500 python tools/parity_campaign.py 4 ref bc7,bc1,bc3,bc6h > $OUT/parity_campaign_4Mpix_vs_reference_kernel.txt 2>&1; tail -1 $OUT/parity_campaign_4Mpix_vs_reference_kernel.txt
  timeout 900 python tools/gpu_settings_fuzz.py 400 6 > $OUT/gpu_settings_fuzz_400.txt 2>&1; tail -1 $OUT/gpu_settings_fuzz_400.txt
  ;;
tables)
  timeout 600 python tools/profile_table.py 2>&1 | nog > $OUT/preset_table.txt
  timeout 600 python tools/variant_table.py 2>&1 | nog > $OUT/variant_table.txt
  timeout 600 python tools/bc7_path_probe.py slow,basic,alpha_basic,veryfast,alpha_slow 2>&1 | nog > $OUT/bc7_path_probe.txt
  timeout 600 python tools/host_path_timing.py 2>&1 | nog > $OUT/host_pointer_path.txt
  timeout 900 python tools/sliced_timing.py 4096 5 0,4,16 2>/dev/null > $OUT/sliced_timing.jsonl
  timeout 600 python tools/sliced_timing.py 16384 2 0 bc7_basic,bc1,bc6h_slow 2>/dev/null > $OUT/sliced_timing_16384.jsonl
  timeout 900 python tools/ref_caller_timing.py 4096 8,64 > $OUT/reference_caller_timing.jsonl 2>&1
  timeout 600 bash tools/combiner_probe.sh 2>&1 | nog > $OUT/combiner_probe.txt
  timeout 600 python tools/bc13_timing.py 2>&1 | nog > $OUT/bc13_timing.txt
  ORDER_PROFILES=slow,alpha_slow timeout 600 python tools/order_timing.py I3 I2 baboon test_a mixed monkey landscape 2>&1 | nog > $OUT/bc7_order_policy_by_content.txt
  ITW_BC7_PILOT_DEBUG=1 ORDER_HOST=0 timeout 300 python tools/order_timing.py I3 I2 baboon test_a mixed monkey landscape 2>&1 | grep "^bc7 pilot\|^==" | uniq -c > $OUT/bc7_pilot_verdicts_by_content.txt
  ( cd /tmp
    for c in I3 baboon; do
      ORDER_HOST=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/tr_$c -o t -- python $GRAFT_REPO_ROOT/tools/order_timing.py $c > /dev/null 2>&1
      f=$(find $GRAFT_REPO_ROOT/$OUT/tr_$c -name '*kernel_trace*.csv' | head -1)
      echo "== $c (default policy), one call" >> $GRAFT_REPO_ROOT/$OUT/bc7_call_timeline.txt
      python $GRAFT_REPO_ROOT/tools/trace_timeline.py $f bc7_pilot_estimate >> $GRAFT_REPO_ROOT/$OUT/bc7_call_timeline.txt 2>&1
      rm -rf $GRAFT_REPO_ROOT/$OUT/tr_$c
    done )
  ;;
bench)
  timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 400 $OUT/bench_default.json
  timeout 900 python bench.py --size 16384 --scaling strong --steps 5 --warmup 1 --no-formats --no-cpu > $OUT/bench_16384_strong_n1.json 2>> $OUT/bench_default.err
  ;;
profile)
  bash tools/profile_gpu.sh $TAG > gpurun_out/profile_gpu_$TAG.log 2>&1; tail -3 gpurun_out/profile_gpu_$TAG.log
  ;;
long)
  timeout 2400 python tools/parity_campaign.py 16 > $OUT/parity_campaign_16Mpix_all_formats.txt 2>&1; tail -n 1 $OUT/parity_campaign_16Mpix_all_formats.txt
  timeout 1500 python tools/parity_campaign.py 64 oracle bc7 slow,alpha_slow > $OUT/parity_campaign_64Mpix_bc7_slow_profiles.txt 2>&1; tail -n 1 $OUT/parity_campaign_64Mpix_bc7_slow_profiles.txt
  timeout 1500 python tools/gpu_settings_fuzz.py 1200 6 > $OUT/gpu_settings_fuzz_1200.txt 2>&1; tail -n 1 $OUT/gpu_settings_fuzz_1200.txt
  timeout 1500 python tools/gpu_settings_fuzz.py 2400 11 > $OUT/gpu_settings_fuzz_2400_seed11.txt 2>&1; tail -n 1 $OUT/gpu_settings_fuzz_2400_seed11.txt
  CAMPAIGN_SEED=13 timeout 1500 python tools/parity_campaign.py 16 oracle bc7 slow,alpha_slow,basic,alpha_basic,veryfast,alpha_veryfast > $OUT/parity_campaign_16Mpix_bc7_fused_seed13.txt 2>&1; tail -n 1 $OUT/parity_campaign_16Mpix_bc7_fused_seed13.txt
  CAMPAIGN_SEED=7 timeout 1500 python tools/parity_campaign.py 4 ref bc7,bc1,bc3,bc6h > $OUT/parity_campaign_4Mpix_vs_reference_kernel_seed7.txt 2>&1; tail -n 1 $OUT/parity_campaign_4Mpix_vs_reference_kernel_seed7.txt
  ;;
matrix)
  bash tools/gpu_env_matrix.sh > /dev/null 2>&1; cp gpurun_out/env_matrix/result.txt $OUT/env_matrix.txt; grep -c passed $OUT/env_matrix.txt
  ;;
esac
done
ls -la $OUT
