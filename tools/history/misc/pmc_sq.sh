#!/bin/bash
# SQ counters of one workload's kernels (gpurun box).  Usage: tools/pmc_sq.sh <workload> <tag>
set -u
WL=${1:-bc7_slow}; TAG=${2:-sq}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_IFETCH SQ_WAIT_INST_LDS"; do
  n=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --kernel-trace --output-format csv --pmc $grp -d $OUT/raw_$n -o pmc -- python $ROOT/bench.py --workload $WL --no-formats --no-cpu --steps 1 --warmup 1 > /dev/null 2> $OUT/log_$n.txt
  f=$(find $OUT/raw_$n -name '*counter_collection*.csv' | head -1)
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r["Kernel_Name"][:60]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    cnt[(k, r["Counter_Name"])] += 1
for k in agg:
    print(k)
    for c, v in agg[k].items():
        print(f"   {c:24s} {v / cnt[(k, c)]:16.0f}  (avg of {cnt[(k, c)]} dispatches)")
PY
  else echo "no counter csv for $n"; tail -5 $OUT/log_$n.txt
  fi
  rm -rf $OUT/raw_$n
done
