# BC1 / BC3: waves per SIMD sweep.  2 / 3 waves: the product kernel with dynamic LDS padding that caps the workgroups per CU
# (ITW_BC13_LDS_PAD); 5 waves: a build whose register allocation targets 5 (96 VGPRs, a few spills).  p2 = arithmetic only.
cd $GRAFT_REPO_ROOT
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
cp $L /tmp/orig.so
run() { echo "== $1 pad=$2"; ITW_BC13_LDS_PAD=$2 timeout 300 python tools/bc13_timing.py 2>&1 | grep -E "^bc" | paste - - - -; }
run orig-4waves 0; run orig-3waves 36864; run orig-2waves 49152; run orig-1wave 65536
cp gpurun_variants/lib_bc1w5.so $L; run w5-5waves 0
cp gpurun_variants/lib_bc1p2.so $L; run p2-4waves 0; run p2-3waves 36864; run p2-2waves 49152; run p2-1wave 65536
cp gpurun_variants/lib_bc1p2w5.so $L; run p2w5-5waves 0
cp /tmp/orig.so $L
