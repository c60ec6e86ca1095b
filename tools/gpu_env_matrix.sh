# the BC7 parity tests under each launch-shape switch (the defaults are what tests -m gpu runs)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/env_matrix
for e in ITW_BC7_BOUND=0 ITW_BC7_FUSED=0 ITW_BC7_ALPHA_PRUNE=0 ITW_COALESCE=0 ITW_BC7_PATH=deep ITW_BC7_PATH=wide ITW_BC6H_PATH=wide \
         ITW_BC7_PILOT_THR=-1 ITW_BC7_PILOT_THR=0 ITW_BC7_PILOT_THR=100 ITW_BC7_BANDS=1 ITW_BC7_COMPACT=0 ITW_STAGED_BANDS=0 ITW_STAGED_VERDICT_THR=0 \
         ITW_SLICE_WINDOW=1 ITW_SLICE_WINDOW=3 ITW_SLICED_PIPELINE=0 ITW_HOST_WINDOWS_OFF=1; do
  echo "== $e"
  env $e timeout 900 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_parity_bc6h.py tests/test_gpu_bc7_paths.py tests/test_gpu_bc7_bound.py tests/test_gpu_vs_reference_kernel.py tests/test_dispatch_layer.py tests/test_gpu_host_pointer_runs.py -m gpu -x -q 2>&1 | tail -2
done | tee gpurun_out/env_matrix/result.txt
