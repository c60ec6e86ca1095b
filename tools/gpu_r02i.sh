cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02i
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02i/pytest_gpu.log 2>&1; tail -3 gpurun_out/r02i/pytest_gpu.log
python tools/host_path_timing.py 2>&1 | grep -v amdgpu | tee gpurun_out/r02i/host_path.txt
timeout 900 python tools/ref_caller_timing.py 4096 8,64 2>&1 | tee gpurun_out/r02i/ref_caller.jsonl | grep -E "BC7_slow|BC7_basic|BC1\"|BC6H_slow"
