cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02k/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02k/pytest_gpu.log
