cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02g
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02g/pytest_gpu.log 2>&1; tail -4 gpurun_out/r02g/pytest_gpu.log
timeout 600 python tools/host_path_timing.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02g/host_path.txt
for runs in "0.125,0.625" "0.0625,0.5" "0.125,0.45,0.775" "0.25,0.65"; do echo "ITW_HOST_RUNS=$runs"; ITW_HOST_RUNS=$runs timeout 600 python tools/host_path_timing.py 2>&1 | grep -E "bc7|bc6h"; done | tee -a gpurun_out/r02g/host_path.txt
timeout 900 python tools/ref_caller_timing.py 4096 8,64 2>&1 | grep -E "BC7_slow|BC7_basic|BC6H" | tee gpurun_out/r02g/ref_caller.jsonl
