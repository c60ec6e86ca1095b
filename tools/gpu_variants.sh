cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/variants
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
for v in orig $(ls gpurun_variants | sed 's/lib_//;s/\.so//'); do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp gpurun_variants/lib_$v.so $L; fi
  echo "== $v"
  timeout 300 python tools/profile_table.py 2>&1 | grep -E "^bc7 +(basic|slow|alpha_slow|fast) "
done | tee gpurun_out/variants/table.txt
cp /tmp/orig.so $L
