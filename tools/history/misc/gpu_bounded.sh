# the bounded BC7 mode order: its own tests, BC7 parity, then timing with the order on / off
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/bounded
timeout 900 python -m pytest tests/test_gpu_bc7_bound.py -m gpu -x -q 2>&1 | tail -5
timeout 1500 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py -m gpu -x -q 2>&1 | tail -5
for e in 1 0; do ITW_BC7_BOUND=$e timeout 600 python tools/bc7_bounded_order_timing.py 2>&1 | grep -v amdgpu; done | tee gpurun_out/bounded/timing.txt
