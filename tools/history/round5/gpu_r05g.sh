# round 5, batch g: staged runs (shape of run 0 from the cached verdict, probe beside a wide run 0); the reference's own callers
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_host_pointer_runs.py tests/test_gpu_bc7_bound.py tests/test_dispatch_layer.py tests/test_example_host.py -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
T="timeout 300 python tools/round5/order_timing.py"
{
  $T I3 I2 baboon test_a mixed monkey landscape
  ITW_STAGED_VERDICT_THR=100 $T I3 baboon
  ITW_STAGED_VERDICT_THR=0 $T I3 baboon
  ORDER_PROFILES=basic,alpha_slow $T I3 baboon
} 2>&1 | grep -v amdgpu.ids | tee $O/order_timing.txt

