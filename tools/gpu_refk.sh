cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/refk
timeout 900 python -m pytest tests/test_gpu_vs_reference_kernel.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/refk/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -16 | tee gpurun_out/refk/smoke.txt
