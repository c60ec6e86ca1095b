# round 5, batch k: two bands for every other fused BC7 profile (reference order: basic, fast, veryfast, ITW_BC7_BOUND=0)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py tests/test_gpu_bc7_bound.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.txt
{
  echo "== default (two bands)"; timeout 300 python tools/profile_table.py 2>&1 | grep "bc7"
  echo "== ITW_BC7_BANDS=1"; ITW_BC7_BANDS=1 timeout 300 python tools/profile_table.py 2>&1 | grep "bc7"
  ORDER_HOST=0 ORDER_PROFILES=basic,fast ITW_BC7_BOUND=0 timeout 300 python tools/round5/order_timing.py I3 baboon
  ORDER_HOST=0 ORDER_PROFILES=slow ITW_BC7_BOUND=0 timeout 300 python tools/round5/order_timing.py I3 baboon
} 2>&1 | grep -v amdgpu.ids | tee $O/preset_tables.txt
