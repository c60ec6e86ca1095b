cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02k
( time timeout 900 python bench.py ) > gpurun_out/r02k/bench_default.json 2> gpurun_out/r02k/bench_default.err; tail -3 gpurun_out/r02k/bench_default.err; python3 -c "
import json
j=json.loads(open('gpurun_out/r02k/bench_default.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('value','ms_per_step','scaling','n_gpus','steps')}); print(j['config']['workload']); print(j['roofline'].get('valu_algorithmic')); print(j['cpu_baseline'])
for k,v in j['formats'].items(): print(k, v)
"
( time timeout 900 python bench.py --size 16384 --scaling strong --steps 3 --warmup 1 --no-formats --no-cpu ) > gpurun_out/r02k/bench_16384_n1.json 2> gpurun_out/r02k/bench_16384.err; tail -3 gpurun_out/r02k/bench_16384.err; tail -c 900 gpurun_out/r02k/bench_16384_n1.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 3 --warmup 1 --no-formats --no-cpu 2>&1 | tail -2 | cut -c1-400
