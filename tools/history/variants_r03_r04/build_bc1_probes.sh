#!/bin/bash
# Builds the BC1 / BC3 measurement variants of DESIGN.md 3.1 in the build container (hipcc cross-compiles gfx950 without a GPU):
#   gpurun_variants/lib_bc1<name>.so = the product library with bc1_bc3.o replaced by tools/variants/bc1_bc3_r03_probes.hip compiled
#   with one switch.  tools/gpu_probe_bc1.sh then times each on the GPU box.  The product build never sees these sources.
set -e
ROOT=$(cd $(dirname $0)/../.. && pwd)
CS=$ROOT/intel-texture-works-plugin_amd/csrc
make -C $CS -j8 > /dev/null
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-function"
mkdir -p $ROOT/gpurun_variants /tmp/bc1_variants
OTHERS=$(ls $CS/build/*.o | grep -v bc1_bc3.o)
build() {   # name, defines...
  local name=$1; shift
  /opt/rocm/bin/hipcc $FL "$@" -c $ROOT/tools/variants/bc1_bc3_r03_probes.hip -o /tmp/bc1_variants/$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/gpurun_variants/lib_bc1$name.so /tmp/bc1_variants/$name.o $OTHERS
  echo "built gpurun_variants/lib_bc1$name.so"
}
build r03                                   # the round-3 kernel with its launch knob (ITW_BC13_LDS_PAD caps the workgroups per CU)
build memonly   -DITW_BC1_PROBE=1           # loads + table staging + stores, no encode
build aluonly   -DITW_BC1_PROBE=2           # the arithmetic alone, no global loads
build loadstore -DITW_BC1_PROBE=3           # loads + stores only
build nopk      -DITW_BC1_PK=0
build nofq      -DITW_BC1_FQUANT=0
build alloff    -DITW_BC1_PK=0 -DITW_BC1_FQUANT=0 -DITW_BC1_ASMCVT=0 -DITW_BC1_EARLYLOAD=0
build w5        -DITW_BC1_WAVES=5
