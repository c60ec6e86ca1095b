# round 5, batch a: parity of the new paths, then the pilot / deep-split / staged-run / compact-list timings
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $O/pytest_gpu.txt
T="timeout 300 python tools/round5/order_timing.py"
{
  $T I3 I2 baboon test_a mixed                                       # defaults: pilot thr 75, compact lists, staged runs wide
  ITW_BC7_PILOT_THR=-1 $T I3 I2 baboon test_a mixed                  # round 4: whole call bounded
  ITW_BC7_PILOT_THR=0 $T I3 I2 baboon test_a                         # pilot, rest always in the reference's order
  ITW_BC7_PILOT_THR=100 $T I3 I2 baboon                              # pilot, rest always bounded
  ITW_BC7_BOUND=0 $T I3 I2 baboon test_a                             # the reference's order, one launch pair
  ITW_BC7_COMPACT=0 ITW_BC7_PILOT_THR=-1 $T I3 baboon                # round 4 exactly (gathering list scans)
  ITW_BC7_PILOT_THR=-1 ITW_BC7_DEEP_SPLIT=2 ORDER_HOST=0 $T I3 baboon
  ITW_BC7_PILOT_THR=-1 ITW_BC7_DEEP_SPLIT=4 ORDER_HOST=0 $T I3 baboon
  ITW_BC7_BOUND=0 ITW_BC7_DEEP_SPLIT=2 ORDER_HOST=0 $T I3 baboon
  ITW_STAGED_WIDE_MAX=1 $T I3 baboon                                 # staged runs of host-pointer calls take the deep shape (pilot on)
  ITW_STAGED_WIDE_MAX=1 ITW_HOST_RUNS=0.125,0.5625 $T I3 baboon
  ITW_STAGED_WIDE_MAX=1 ITW_HOST_RUNS=0.25,0.625 $T I3 baboon
  ITW_STAGED_WIDE_MAX=1 ITW_HOST_RUNS=0.0625,0.375,0.6875 $T I3 baboon
  ITW_STAGED_WIDE_MAX=1 ITW_BC7_PILOT_THR=-1 $T I3 baboon
  ORDER_PROFILES=alpha_slow ORDER_HOST=0 $T I3 baboon
} 2>&1 | grep -v amdgpu.ids | tee $O/order_timing.txt
# per-kernel times of the default policy on three contents
cd /tmp
for c in I3 I2 baboon; do
  ORDER_HOST=0 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$c -o p -- python $GRAFT_REPO_ROOT/tools/round5/order_timing.py $c > /dev/null 2>&1
  f=$(find $GRAFT_REPO_ROOT/$O/prof_$c -name '*kernel_stats*.csv' | head -1)
  echo "== $c" >> $GRAFT_REPO_ROOT/$O/kernel_times.txt
  [ -n "$f" ] && python - "$f" >> $GRAFT_REPO_ROOT/$O/kernel_times.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "bc7" in r["Name"]:
        print(f'{r["Name"][:78]:78s} calls {int(r["Calls"]):4d} avg {float(r["AverageNs"])/1e6:8.3f} ms')
PY
  rm -rf $GRAFT_REPO_ROOT/$O/prof_$c
done
cat $GRAFT_REPO_ROOT/$O/kernel_times.txt
