cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02j/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02j/pytest_gpu.log
timeout 600 python tools/profile_table.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02j/profile_table.txt
