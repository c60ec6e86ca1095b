# A/B of gpurun_variants/lib_bc1*.so: BC1 / BC3 timing (tools/bc13_timing.py) + parity of the variant
cd $GRAFT_REPO_ROOT
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
for v in orig $(ls gpurun_variants | grep bc1 | sed 's/lib_//;s/\.so//') orig; do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp gpurun_variants/lib_$v.so $L; fi
  echo "== $v"
  timeout 300 python tools/bc13_timing.py 2>&1 | grep -E "^bc" | paste - - - -
  if [ $v != orig ]; then timeout 600 python -m pytest tests/test_gpu_parity_bc1_bc3.py -m gpu -q -x 2>&1 | tail -1; fi
done
cp /tmp/orig.so $L
