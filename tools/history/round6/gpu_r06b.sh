#!/bin/bash
# round 6, batch b: where the time of the presets the plugin selects goes (kernel trace + SQ counters per preset)
set -u
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r06b
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in bc7_basic bc7_veryfast bc7_alpha_basic bc7_alpha_veryfast bc6h_fast; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$wl -o trace -- python $ROOT/bench.py --workload $wl --no-formats --no-cpu --steps 5 --warmup 1 > $OUT/bench_$wl.json 2> $OUT/trace_$wl.log
  find $OUT/trace_$wl -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_$wl.csv \;
  rm -rf $OUT/trace_$wl
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- python $ROOT/bench.py --workload $wl --no-formats --no-cpu --steps 2 --warmup 1 > $OUT/pmc_sq_$wl.json 2> $OUT/pmc_sq_$wl.log
  f=$(find $OUT/pmc_sq -name '*counter_collection*.csv' | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/pmc_sq_$wl.csv; grep -E 'bc7_|bc6h_' $f | head -6000 >> $OUT/pmc_sq_$wl.csv; fi
  rm -rf $OUT/pmc_sq
  echo "== $wl"; cut -d, -f1-4,8 $OUT/kernel_stats_$wl.csv | head -8
done
