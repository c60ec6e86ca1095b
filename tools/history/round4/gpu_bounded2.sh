cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bc7_bound.py -m gpu -x -q 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_bounded_variants.sh
