# A/B of every gpurun_variants/lib_*.so WITH parity: BC7 parity suites under each library, then the preset table rows
cd $GRAFT_REPO_ROOT
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
for v in orig $(ls gpurun_variants | sed 's/lib_//;s/\.so//'); do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp gpurun_variants/lib_$v.so $L; fi
  echo "== $v"
  timeout 900 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py -m gpu -q -x 2>&1 | tail -1
  timeout 300 python tools/profile_table.py 2>&1 | grep -E "^bc7 "
done
cp /tmp/orig.so $L
