set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest_gpu.log
tail -5 gpurun_out/r02a/pytest_gpu.log
timeout 600 python bench.py > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; tail -c 1500 gpurun_out/r02a/bench.json
timeout 900 python tools/ref_caller_timing.py 4096 8,64 > gpurun_out/r02a/ref_caller_timing.jsonl 2>&1; cat gpurun_out/r02a/ref_caller_timing.jsonl
nproc; lscpu | head -20
