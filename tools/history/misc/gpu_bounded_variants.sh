# A/B of gpurun_variants/lib_*.so builds of the bounded BC7 order (bail-out thresholds, bound tightness, split cost alone)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bounded
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
for v in orig $(ls gpurun_variants | sed 's/lib_//;s/\.so//'); do
  if [ $v = orig ]; then cp /tmp/orig.so $L; else cp gpurun_variants/lib_$v.so $L; fi
  echo "== $v"
  BOUNDED_QUICK=1 timeout 300 python tools/bc7_bounded_order_timing.py 2>&1 | grep -E "slow"
done | tee gpurun_out/bounded/variants.txt
cp /tmp/orig.so $L
echo "== classic order"; ITW_BC7_BOUND=0 BOUNDED_QUICK=1 timeout 300 python tools/bc7_bounded_order_timing.py 2>&1 | grep -E "slow" | tee -a gpurun_out/bounded/variants.txt
