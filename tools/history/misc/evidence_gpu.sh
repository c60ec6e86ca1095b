#!/bin/bash
# Round evidence, one gpurun call:  tools/evidence_gpu.sh <tag>   -> gpurun_out/evidence_<tag>/ (+ gpurun_out/prof_<tag>/)
# Then in the build container: python tools/summarize_profiles.py <tag>; cp gpurun_out/evidence_<tag>/* profiles/ (named <tag>_*).
TAG=${1:-r03}
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/evidence_$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
timeout 900 python tools/parity_campaign.py 2 > $OUT/parity_campaign_2Mpix_wide.txt 2>&1; tail -1 $OUT/parity_campaign_2Mpix_wide.txt
timeout 1800 python tools/parity_campaign.py 16 > $OUT/parity_campaign_16Mpix_fused.txt 2>&1; tail -1 $OUT/parity_campaign_16Mpix_fused.txt
timeout 1800 python tools/parity_campaign.py 4 ref > $OUT/parity_campaign_4Mpix_vs_reference_kernel.txt 2>&1; tail -1 $OUT/parity_campaign_4Mpix_vs_reference_kernel.txt
timeout 600 python tools/profile_table.py 2>&1 | grep -v amdgpu > $OUT/preset_table.txt
timeout 600 python tools/bc7_path_probe.py slow,basic,alpha_basic,veryfast,alpha_slow 2>&1 | grep -v amdgpu > $OUT/bc7_path_probe.txt
timeout 600 python tools/bc7_path_probe.py slow,veryslow bc6h 2>&1 | grep -v amdgpu > $OUT/bc6h_path_probe.txt
timeout 600 python tools/host_path_timing.py 2>&1 | grep -v amdgpu > $OUT/host_pointer_path.txt
timeout 900 python tools/ref_caller_timing.py 4096 8,64 > $OUT/reference_caller_timing.jsonl 2>&1
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 900 python bench.py --size 16384 --scaling strong --steps 5 --warmup 1 --no-formats --no-cpu > $OUT/bench_16384_strong_n1.json 2>> $OUT/bench_default.err
bash tools/profile_gpu.sh $TAG > $OUT/profile_gpu.log 2>&1
bash tools/gpu_traffic.sh > $OUT/bc7_traffic_by_shape.txt 2>&1
ls -la $OUT
