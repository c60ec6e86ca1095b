cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/evidence_r02c
timeout 1800 python tools/parity_campaign.py 4 ref 2>&1 | grep -v amdgpu | tee gpurun_out/evidence_r02c/parity_campaign_4Mpix_vs_reference_kernel.txt | tail -22
