# Where does a BC1 / BC3 launch spend its time?    Build variants of tools/variants/bc1_bc3_r03_probes.hip (the
# round-3 kernel with its A/B switches and probe builds; the shipped csrc/bc1_bc3.hip has one code path), made in the container by
# tools/variants/build_bc1_probes.sh into gpurun_variants/lib_bc1<name>.so:
#   r02        the round-2 kernel                      memonly    loads + table staging + stores, no encode  (ITW_BC1_PROBE=1)
#   loadstore  loads + stores only                     aluonly    the arithmetic alone, no global loads      (ITW_BC1_PROBE=2)
#   nopk / nofq / alloff   the round-3 instruction-level changes switched off one by one / all
#   w5, aluonly_w5         register allocation for 5 waves per SIMD (96 VGPRs, a few spills)
# and the r03 variant under ITW_BC13_LDS_PAD (dynamic LDS that caps the workgroups per CU: 3, 2 waves per SIMD; the knob exists
# in the variant builds only).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe_bc1
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
run() { echo "== $1"; ITW_BC13_LDS_PAD=${2:-0} timeout 300 python tools/bc13_timing.py 2>&1 | grep -E "^bc" | paste - - - -; }
{
run "product (4 waves per SIMD)"
if [ -f gpurun_variants/lib_bc1r03.so ]; then
  cp gpurun_variants/lib_bc1r03.so $L
  run "r03 variant, 3 waves per SIMD (LDS pad 36 KiB)" 36864
  run "r03 variant, 2 waves per SIMD (LDS pad 48 KiB)" 49152
fi
for v in $(ls gpurun_variants | grep '^lib_bc1' | sed 's/lib_bc1//;s/\.so//'); do
  cp gpurun_variants/lib_bc1$v.so $L
  run "$v"
  case $v in aluonly) run "aluonly, 2 waves per SIMD" 49152;; esac
done
} 2>&1 | tee gpurun_out/probe_bc1/table.txt
cp /tmp/orig.so $L
