# round 5, batch i: bands for the bounded alpha_slow chain
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bc7_bound.py tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py tests/test_gpu_host_pointer_runs.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
T="timeout 300 python tools/round5/order_timing.py"
{
  ORDER_PROFILES=alpha_slow $T I3 baboon mixed
  ORDER_PROFILES=alpha_slow ITW_BC7_BANDS=1 $T I3 baboon mixed
  ORDER_PROFILES=alpha_slow ITW_BC7_BOUND=0 ORDER_HOST=0 $T I3 baboon
} 2>&1 | grep -v amdgpu.ids | tee $O/order_timing.txt
