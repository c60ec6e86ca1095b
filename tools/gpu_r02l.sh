# refresh of the all-formats kernel stats (without the 16384^2 side figures) + BC1/BC3 at both sizes
cd $GRAFT_REPO_ROOT
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_r02c; mkdir -p $OUT
t() { timeout 300 python bench.py --workload $2 --no-formats --no-cpu --steps 300 --warmup 30 --size $3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1', j['ms_per_step'], 'ms/step; kernel avg', r['kernel_ms_avg'], 'min', r['kernel_ms_min'], 'frac', r['frac'])"; }
( t "bc1 4096" bc1 4096; t "bc3 4096" bc3 4096; t "bc1 16384" bc1 16384; t "bc3 16384" bc3 16384; t "bc1 4096" bc1 4096; t "bc3 4096" bc3 4096 ) | tee $ROOT/gpurun_out/evidence_r02c/bc1_bc3_sizes.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --no-cpu --no-16k > $OUT/bench_all_formats_under_rocprof.json 2> $OUT/trace_all.log
find $OUT/trace -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_all_formats.csv \;
rm -rf $OUT/trace
cat $OUT/kernel_stats_all_formats.csv | cut -c1-60,150-260
