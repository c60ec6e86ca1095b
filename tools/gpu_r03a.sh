# round 3, first GPU pass: parity of the reworked BC7 palette path / BC1-BC3 quantisation, the new multi-GPU tests, timings
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03a; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; tail -5 $OUT/pytest_gpu.log
timeout 600 python tools/profile_table.py 2>&1 | grep -v amdgpu > $OUT/preset_table.txt; cat $OUT/preset_table.txt
timeout 900 python bench.py --no-cpu > $OUT/bench_default.json 2> $OUT/bench_default.err; tail -c 6000 $OUT/bench_default.json
