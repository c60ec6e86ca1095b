# A/B of gpurun_variants/lib_bc45*.so: BC4 / BC5 timing (tools/profile_table.py rows)
cd $GRAFT_REPO_ROOT
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
for v in orig $(ls gpurun_variants | grep bc45 | sed 's/lib_//;s/\.so//'); do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp gpurun_variants/lib_$v.so $L; fi
  echo "== $v"
  for i in 1 2; do timeout 300 python tools/profile_table.py 2>&1 | grep -E "^bc[45] "; done
done
cp /tmp/orig.so $L
