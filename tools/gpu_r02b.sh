set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02b
timeout 900 python -m pytest tests/test_gpu_bc7_paths.py -x -q > gpurun_out/r02b/pytest_paths.log 2>&1; tail -15 gpurun_out/r02b/pytest_paths.log
timeout 900 python tools/bc7_path_probe.py > gpurun_out/r02b/path_probe.txt 2>&1; cat gpurun_out/r02b/path_probe.txt
