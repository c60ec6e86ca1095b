cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
timeout 900 python -m pytest tests/test_gpu_bc7_paths.py tests/test_gpu_parity_bc7.py -x -q 2>&1 | tail -3
timeout 900 python tools/bc7_path_probe.py slow,basic,alpha_basic 2>&1 | grep -v amdgpu.ids
bash tools/gpu_r02e.sh 2>&1 | head -8
timeout 900 python tools/ref_caller_timing.py 4096 64 2>&1 | grep -E "BC7_slow|BC7_basic|BC7_alpha" | grep '"slices": 64'
