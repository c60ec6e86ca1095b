# round 5, batch c: pilot on a third (priority) stream beside two bands; staged host-pointer runs as overlapped deep bands
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bc7_bound.py tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py tests/test_dispatch_layer.py tests/test_example_host.py tests/test_gpu_multigpu_cpp.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
T="timeout 300 python tools/round5/order_timing.py"
{
  $T I3 I2 baboon test_a mixed
  ITW_BC7_PILOT_THR=-1 $T I3 I2 baboon test_a
  ITW_BC7_PILOT_THR=0 ORDER_HOST=0 $T I3 baboon
  ITW_BC7_PILOT_THR=100 ORDER_HOST=0 $T I3 baboon
  ITW_STAGED_BANDS=0 $T I3 baboon
  ITW_HOST_RUNS=0.125,0.5625 $T I3 baboon
  ITW_HOST_RUNS=0.0625,0.5 $T I3 baboon
  ITW_HOST_RUNS=0.125,0.4375,0.75 $T I3 baboon
  ITW_HOST_RUNS=0.25,0.625 $T I3 baboon
  ITW_HOST_RUNS=0.5 $T I3 baboon
  ORDER_PROFILES=basic,alpha_slow $T I3 baboon
  ITW_STAGED_BANDS=0 ORDER_PROFILES=basic,alpha_slow $T I3 baboon
} 2>&1 | grep -v amdgpu.ids | tee $O/order_timing.txt
