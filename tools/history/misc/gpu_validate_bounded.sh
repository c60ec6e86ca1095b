# after the bounded BC7 order: env matrix, settings fuzz (a third of the trials in the bounded order's territory), big campaign of the two profiles
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/validate
timeout 1200 python tools/gpu_settings_fuzz.py 1200 2027 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/validate/gpu_settings_fuzz_1200.txt
timeout 1800 python tools/parity_campaign.py 64 oracle bc7 slow,alpha_slow 2>&1 | grep -v amdgpu | tee gpurun_out/validate/parity_campaign_64Mpix_bounded_order.txt
bash tools/gpu_env_matrix.sh 2>&1 | tee gpurun_out/validate/env_matrix.txt
