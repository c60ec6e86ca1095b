# last GPU action after a late kernel change: the BC7 suites, then the rocprofv3 passes (stamps) -- tools/evidence_gpu.sh's other outputs stay
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests/test_gpu_bc7_bound.py tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py tests/test_gpu_bench_contract.py -m gpu -q 2>&1 | tail -3 | tee gpurun_out/final/pytest_bc7.txt
timeout 600 python tools/parity_campaign.py 8 oracle bc7 slow,alpha_slow 2>&1 | grep -v amdgpu | tee gpurun_out/final/parity_campaign_8Mpix_slow_profiles.txt
bash tools/profile_gpu.sh r04 > gpurun_out/final/profile_gpu.log 2>&1
tail -3 gpurun_out/final/profile_gpu.log
