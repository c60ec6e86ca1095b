#!/bin/bash
# Round 4: staggered wide schedule -- parity of both launch shapes + timing by call size + the reference's own callers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bc7_paths.py tests/test_gpu_parity_bc7.py tests/test_dispatch_layer.py -m gpu -q -x 2>&1 | tail -2
timeout 600 python tools/bc7_path_probe.py slow,alpha_slow,basic,alpha_basic 2>&1 | grep -v amdgpu | tee $O/bc7_path_probe.txt
timeout 900 python tools/ref_caller_timing.py 4096 8,64 > $O/reference_caller_timing.jsonl 2>&1; cut -c1-400 $O/reference_caller_timing.jsonl | head -20
