#!/bin/bash
# Round 4, third GPU pass: BC4/BC5 with batched table reads; prune-rate probe of the BC7 scans (tools/variants/make_bc7_prune_probe.py)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity_bc4_bc5.py tests/test_reference_codecs.py -m gpu -q -x > $O/pytest_bc45.log 2>&1; tail -2 $O/pytest_bc45.log
timeout 600 python tools/parity_campaign.py 8 oracle bc4,bc5 > $O/parity_campaign_bc45_8Mpix.txt 2>&1; tail -1 $O/parity_campaign_bc45_8Mpix.txt
for i in 1 2; do timeout 300 python tools/profile_table.py 2>&1 | grep -E "^bc[1345] "; done | tee $O/preset_table_bc1345.txt
L=intel-texture-works-plugin_amd/lib/libispc_texcomp.so
if [ -f gpurun_variants/lib_pruneprobe.so ]; then
  cp $L /tmp/orig.so; cp gpurun_variants/lib_pruneprobe.so $L
  timeout 600 python tools/variants/bc7_prune_probe_run.py 2>&1 | grep -v amdgpu | tee $O/bc7_prune_probe.txt
  cp /tmp/orig.so $L
fi
