# round 5, batch j: smoke(); the C++ multi-GPU bench job with 8 virtual ranks on the one device, K = 4 (default), 2, 1
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -8 | tee $O/smoke.txt
for k in 4 2 1; do
  ITW_MULTIGPU_INTERLEAVE=$k ITW_BENCH_CPP_SHARE_DEVICES=1 timeout 600 python bench.py --cpp-worker --gpus 8 --steps 3 --warmup 1 --size 16384 --workload bc7_slow 2> /dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('K =', j.get('sub_bands_per_rank'), 'ms_per_step', j['ms_per_step'], 'verified', j['gather_verified'], 'checks', j['band_checks'], 'transport', j['transport'], 'scatter', (j.get('scatter_from_gpu0') or {}).get('ms_per_step'))" | tee -a $O/cpp_worker_8virtual.txt
done
