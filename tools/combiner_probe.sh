#!/bin/bash
# What the call combiner does under the reference's own pool (win32Threads.cpp compiled unmodified, oracle/_ref/ref_threads_caller_gpu) with 8 / 16 / 64
# pool threads: bursts, leader rounds, requests and merged calls per run (ITW_COALESCE_DEBUG=1), next to the time of the fastest of 4 passes.
# "the oversubscribed 64-thread case is not analysed".  Run by tools/evidence.sh (tables).
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys, os
sys.path.insert(0, '.'); sys.path.insert(0, 'intel-texture-works-plugin_amd')
from itw_amd import surfaces
surfaces.ldr_smooth(4096, 4096).tofile('/tmp/ldr.raw')
PY
echo "usable cores: $(python -c 'import os; print(len(os.sched_getaffinity(0)))') nproc $(nproc)"
for tramp in BC7_basic BC1; do
for w in 8 16 64; do
  for shape in whole slices; do
    extra=""; [ $shape = whole ] && extra=whole
    echo "== $tramp workers $w $shape"
    ITW_COALESCE_DEBUG=1 ITW_REF_THREADS=$w ITW_REF_REPS=4 oracle/_ref/ref_threads_caller_gpu mt $tramp 4096 4096 /tmp/ldr.raw /tmp/out.bin $extra 2>&1 | tail -2
  done
done
done
