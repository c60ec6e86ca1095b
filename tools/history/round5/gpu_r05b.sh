# round 5, batch b: pilot + two bands in the fused `slow` path; staged host-pointer runs through it
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bc7_bound.py tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py tests/test_abi_boundary.py tests/test_dispatch_layer.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
T="timeout 300 python tools/round5/order_timing.py"
{
  $T I3 I2 baboon test_a mixed
  ITW_BC7_PILOT_THR=-1 ORDER_HOST=0 $T I3 I2 baboon
  ITW_BC7_BANDS=1 ORDER_HOST=0 $T I3 baboon
  ITW_BC7_PILOT_THR=0 ORDER_HOST=0 $T I3 baboon
  ITW_BC7_PILOT_THR=100 ORDER_HOST=0 $T I3 baboon
  ITW_STAGED_WIDE_MAX=1 $T I3 I2 baboon
  ITW_STAGED_WIDE_MAX=1 ITW_HOST_RUNS=0.125,0.5625 $T I3 baboon
  ITW_STAGED_WIDE_MAX=1 ITW_HOST_RUNS=0.1875,0.625 $T I3 baboon
  ITW_STAGED_WIDE_MAX=1 ITW_HOST_RUNS=0.25 $T I3 baboon
  ITW_STAGED_WIDE_MAX=1 ITW_HOST_CHUNKS=1 $T I3 baboon
  ORDER_SIZE=16384 ORDER_HOST=0 $T I3
  ORDER_SIZE=16384 ORDER_HOST=0 ITW_BC7_BANDS=1 $T I3
} 2>&1 | grep -v amdgpu.ids | tee $O/order_timing.txt
