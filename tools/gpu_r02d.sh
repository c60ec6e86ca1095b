cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02d
timeout 900 python -m pytest tests/test_dispatch_layer.py tests/test_reference_pins.py tests/test_gpu_parity_bc1_bc3.py -m gpu -x -q > gpurun_out/r02d/pytest.log 2>&1; tail -5 gpurun_out/r02d/pytest.log
timeout 900 python tools/ref_caller_timing.py 4096 8,64 > gpurun_out/r02d/ref_caller_timing.jsonl 2>&1; cat gpurun_out/r02d/ref_caller_timing.jsonl
