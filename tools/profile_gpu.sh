#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats of the default bench command, then separate PMC
# passes (FETCH_SIZE / WRITE_SIZE need separate passes: TCC has 4 slots, FETCH_SIZE takes 3, WRITE_SIZE 2), then SQ
# counters of the BC7 kernels.  Usage: tools/profile_gpu.sh <tag>   -> gpurun_out/prof_<tag>/...
# Afterwards (in the build container): python tools/summarize_profiles.py <tag>  copies the summaries into profiles/.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
# every summary of this pass is stamped with the kernel sources it was taken on (bench.py drops committed counters whose stamp
# differs from the tree it runs from: VERDICT r03 item 6)
python -c "import sys; sys.path.insert(0, '$ROOT/intel-texture-works-plugin_amd'); import itw_amd; print(itw_amd.source_sha256())" > $OUT/source_sha256.txt
cd /tmp && export TMPDIR=/tmp
KERNELS='bc7_|bc13_kernel|bc6h_|bc45_kernel'

# 1. kernel trace + stats of the headline workload alone (the timed region of the default bench command: BC7 slow), so the
#    per-kernel averages are those of one workload; then of the whole default command (all side formats)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --no-formats --no-cpu > $OUT/bench_under_rocprof.json 2> $OUT/trace.log
find $OUT/trace -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/trace -name '*kernel_trace*.csv' | head -1 | xargs -I{} sh -c "head -1 {} > $OUT/kernel_trace_head.csv; grep -m10 bc7_ {} >> $OUT/kernel_trace_head.csv"
rm -rf $OUT/trace
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python $ROOT/bench.py --no-cpu --no-16k > $OUT/bench_all_formats_under_rocprof.json 2> $OUT/trace_all.log
find $OUT/trace -name '*kernel_stats*.csv' -exec cp {} $OUT/kernel_stats_all_formats.csv \;
find $OUT/trace -name '*kernel_trace*.csv' | head -1 | xargs -I{} sh -c "grep -m3 bc13_kernel {} >> $OUT/kernel_trace_head.csv; grep -m3 bc6h_ {} >> $OUT/kernel_trace_head.csv; grep -m3 bc45_ {} >> $OUT/kernel_trace_head.csv"
rm -rf $OUT/trace

# 2. PMC passes, one counter group per run, short workloads.  The bench line of every pass is kept beside its counter file: its `abi_calls` is
#    the number of C-ABI calls the counters are divided by (tools/summarize_profiles.py; never inferred from dispatch counts).
#    The presets the plugin selects (IntelPlugin.cpp:832-843: veryfast / basic / alpha_veryfast / alpha_basic, BC6H fast / slow) are profiled like the slow ones.
WLS=${ITW_PROFILE_WORKLOADS:-"bc7_slow bc7_alpha_slow bc7_basic bc7_veryfast bc7_alpha_basic bc7_alpha_veryfast bc6h_slow bc6h_fast bc1 bc3 bc4 bc5"}
for wl in $WLS; do
  steps=3; case $wl in bc1|bc3|bc4|bc5) steps=10;; esac
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${wl}_$ctr -o pmc -- python $ROOT/bench.py --workload $wl --no-formats --no-cpu --steps $steps --warmup 1 > $OUT/pmc_${wl}_$ctr.json 2> $OUT/pmc_${wl}_$ctr.log
    f=$(find $OUT/pmc_${wl}_$ctr -name '*counter_collection*.csv' | head -1)
    if [ -n "$f" ]; then head -1 $f > $OUT/pmc_${wl}_$ctr.csv; grep -E "$KERNELS" $f | head -4000 >> $OUT/pmc_${wl}_$ctr.csv; fi
    rm -rf $OUT/pmc_${wl}_$ctr
  done
done
# 3. SQ counters (VALU instruction counts, wave cycles) per workload: the roofline that binds for every format.
#    A --pmc run serialises the dispatches, so this pass also gives every kernel's ISOLATED duration (bench.py: roofline.rocprof_kernels.isolated).
for wl in $WLS; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/pmc_sq -o pmc -- python $ROOT/bench.py --workload $wl --no-formats --no-cpu --steps 2 --warmup 1 > $OUT/pmc_sq_$wl.json 2> $OUT/pmc_sq_$wl.log
  f=$(find $OUT/pmc_sq -name '*counter_collection*.csv' | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/pmc_sq_$wl.csv; grep -E "$KERNELS" $f | head -6000 >> $OUT/pmc_sq_$wl.csv; fi
  rm -rf $OUT/pmc_sq
done
cp $OUT/pmc_sq_bc7_slow.csv $OUT/pmc_sq_bc7.csv 2>/dev/null
ls -la $OUT
