#!/bin/bash
# Round 5 evidence, part C: after the last source edit (host-side only: abi.hip's guard, comments) -- the whole GPU suite, the rocprofv3 passes
# again (their stamps must equal the tree), then the bench lines with those passes' predecessors replaced by tools/summarize_profiles.py afterwards
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/evidence_r05; mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -2 $OUT/pytest_gpu.log
bash tools/profile_gpu.sh r05 > gpurun_out/profile_gpu_r05.log 2>&1; tail -3 gpurun_out/profile_gpu_r05.log
