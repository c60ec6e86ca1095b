cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/evidence_r03
timeout 1500 python tools/gpu_settings_fuzz.py 1500 2026 2>&1 | grep -v amdgpu | tail -3 | tee gpurun_out/evidence_r03/gpu_settings_fuzz_1500.txt
