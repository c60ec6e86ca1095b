# round 5, batch d: host-side verdict of staged runs; timeline of the pilot beside the bands
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_bc7_bound.py tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_dispatch_layer.py tests/test_example_host.py -m gpu -x -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
T="timeout 300 python tools/round5/order_timing.py"
{
  $T I3 I2 baboon test_a mixed
  ITW_STAGED_VERDICT_THR=100 $T I3 baboon
  ITW_STAGED_VERDICT_THR=0 $T I3 baboon
  ITW_STAGED_VERDICT_THR=0 ITW_HOST_RUNS=0.0625,0.5 $T I3 baboon
  ITW_HOST_RUNS=0.0625,0.5 $T I3 baboon
} 2>&1 | grep -v amdgpu.ids | tee $O/order_timing.txt
cd /tmp
for c in I3 baboon; do
  ORDER_HOST=0 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/tr_$c -o t -- python $GRAFT_REPO_ROOT/tools/round5/order_timing.py $c > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/tr_$c -name '*kernel_trace*.csv' | head -1)
  echo "== $c (default policy)" >> $GRAFT_REPO_ROOT/$O/timeline.txt
  python $GRAFT_REPO_ROOT/tools/round5/trace_timeline.py $f >> $GRAFT_REPO_ROOT/$O/timeline.txt 2>&1
  rm -rf $GRAFT_REPO_ROOT/$O/tr_$c
done
cat $GRAFT_REPO_ROOT/$O/timeline.txt
