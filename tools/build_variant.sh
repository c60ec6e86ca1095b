#!/bin/bash
# Builds a kernel variant for A/B timing on the GPU box: tools/build_variant.sh <name> "<extra -D flags>" [source stems, default: bc7]
#   -> gpurun_variants/lib_<name>.so  (the named sources recompiled with the Makefile's flags + the extra ones, linked with the tree's other objects)
# tools/gpu_variants.sh then swaps every gpurun_variants/lib_*.so in turn and prints tools/variant_table.py's rows.
set -e
NAME=$1; EXTRA=$2; shift 2 || true
STEMS=${@:-bc7}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
CS=$ROOT/intel-texture-works-plugin_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -fvisibility=hidden -Wall -Wno-unused-function"
mkdir -p $ROOT/gpurun_variants /tmp/variant_$NAME
make -C $CS -j8 >/dev/null
OBJS=""
for o in $CS/build/*.o; do
  b=$(basename $o .o)
  case $b in *.test) continue;; esac
  skip=0; for s in $STEMS; do [ "$b" = "$s" ] && skip=1; done
  [ $skip = 0 ] && OBJS="$OBJS $o"
done
for s in $STEMS; do
  PER=""; [ "$s" = bc7 ] && [ -z "${NO_PER_SOURCE:-}" ] && PER="-mllvm -amdgpu-sched-strategy=max-ilp"     # the Makefile's EXTRA_bc7 (NO_PER_SOURCE=1: without)
  /opt/rocm/bin/hipcc $FLAGS $PER $EXTRA -c $CS/$s.hip -o /tmp/variant_$NAME/$s.o &
done
wait
for s in $STEMS; do OBJS="$OBJS /tmp/variant_$NAME/$s.o"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,-Bsymbolic-functions -o $ROOT/gpurun_variants/lib_$NAME.so $OBJS
echo built gpurun_variants/lib_$NAME.so
