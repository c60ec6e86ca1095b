cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02f
timeout 900 python -m pytest tests/test_gpu_bc7_paths.py -x -q 2>&1 | tail -3
timeout 900 python tools/bc7_path_probe.py slow,basic,alpha_slow 2>&1 | grep -v amdgpu.ids
bash tools/gpu_r02e.sh 2>&1 | head -12
