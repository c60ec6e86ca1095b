cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bc7_paths.py tests/test_gpu_parity_bc6h.py -x -q 2>&1 | tail -6
timeout 600 python tools/bc7_path_probe.py slow,veryslow bc6h 2>&1 | grep -v amdgpu
