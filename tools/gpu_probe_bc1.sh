# Where does a BC1 / BC3 launch spend its time?  Build variants (gpurun_variants/lib_*.so, built in the container):
#   bc1old     round-2 kernel            bc1probe1  memory side only (loads + table staging + stores, no encode)
#   bc1probe2  arithmetic only (no global loads)            bc1probe3  loads + stores only (no staging, no encode)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/probe_bc1
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
for v in orig $(ls gpurun_variants | grep bc1 | sed 's/lib_//;s/\.so//'); do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp gpurun_variants/lib_$v.so $L; fi
  echo "== $v"
  timeout 300 python tools/bc13_timing.py 2>&1 | grep -E "^bc"
done | tee gpurun_out/probe_bc1/table.txt
cp /tmp/orig.so $L
