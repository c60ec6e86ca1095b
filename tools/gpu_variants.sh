# A/B of kernel build variants on the GPU box: every gpurun_variants/lib_<name>.so (built in the container, e.g.
#   hipcc <Makefile FLAGS> -DSWALL=2 -c csrc/bc7.hip -o /tmp/bc7_sw2.o && hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/lib_sw2.so /tmp/bc7_sw2.o <the other build/*.o>
# ) replaces the library in turn; prints the preset table rows.  Used for DESIGN.md 3.2: scan at 4 / 3 / 2 waves per SIMD.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/variants
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
for v in orig $(ls gpurun_variants | sed 's/lib_//;s/\.so//'); do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp gpurun_variants/lib_$v.so $L; fi
  echo "== $v"
  timeout 900 ${VARIANT_CMD:-python tools/variant_table.py ${VARIANT_FILTER:-}} 2>&1 | grep -E "${VARIANT_GREP:- ms }"
done | tee gpurun_out/variants/table.txt
cp /tmp/orig.so $L
