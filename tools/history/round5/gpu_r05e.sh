# round 5, batch e: striped bands + estimate pilot; staged runs (first run on the second stream, host-side verdict)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
T="timeout 300 python tools/round5/order_timing.py"
{
  $T I3 I2 baboon test_a mixed monkey landscape
  ITW_BC7_PILOT_THR=-1 ORDER_HOST=0 $T I3 I2 baboon test_a
  ITW_BC7_PILOT_THR=0 ORDER_HOST=0 $T I3 I2 baboon test_a
  ITW_BC7_PILOT_THR=100 ORDER_HOST=0 $T I3 I2 baboon test_a
  ITW_BC7_PILOT_DEBUG=1 ORDER_HOST=0 $T I3 I2 baboon test_a mixed monkey landscape 2>&1 | grep "^bc7 pilot\|^==" | uniq -c
  ITW_STAGED_VERDICT_THR=100 $T I3 baboon
  ITW_STAGED_VERDICT_THR=0 $T I3 baboon
  ITW_STAGED_BANDS=0 $T I3 baboon
} 2>&1 | grep -v amdgpu.ids | tee $O/order_timing.txt
