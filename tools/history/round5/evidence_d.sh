#!/bin/bash
# Round 5 evidence, part D: itwSetBc7Pilot(< -1) restores the environment's preset, so a test that forces the pilot no longer leaves the rest of an
# environment-matrix run at the default: the three ITW_BC7_PILOT_THR rows again; then the rocprofv3 passes and the bench lines on these sources
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/env_matrix gpurun_out/r05final
export TMPDIR=/tmp
for e in ITW_BC7_PILOT_THR=-1 ITW_BC7_PILOT_THR=0 ITW_BC7_PILOT_THR=100; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_parity_bc6h.py tests/test_gpu_bc7_paths.py tests/test_gpu_bc7_bound.py tests/test_gpu_vs_reference_kernel.py tests/test_dispatch_layer.py tests/test_gpu_host_pointer_runs.py -m gpu -x -q 2>&1 | tail -2
done | tee gpurun_out/env_matrix/result_pilot_rows.txt
bash tools/profile_gpu.sh r05 > gpurun_out/profile_gpu_r05.log 2>&1; tail -2 gpurun_out/profile_gpu_r05.log
