#!/bin/bash
# Round 5 evidence, part B (one gpurun call): rocprofv3 passes (kernel stats, PMC traffic, SQ counters) and the environment-switch matrix
cd $GRAFT_REPO_ROOT
bash tools/profile_gpu.sh r05 > gpurun_out/profile_gpu_r05.log 2>&1; tail -3 gpurun_out/profile_gpu_r05.log
bash tools/gpu_env_matrix.sh
