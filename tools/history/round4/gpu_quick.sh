# quick check after a BC7 kernel change: parity tests of the BC7 files + the preset table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/quick
timeout 1200 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py tests/test_gpu_vs_reference_kernel.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do timeout 600 python tools/profile_table.py 2>&1 | grep -E "^bc7" ; done | tee gpurun_out/quick/preset_table.txt
