cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity_bc1_bc3.py -x -q 2>&1 | tail -2
t() { timeout 300 python bench.py --workload $2 --no-formats --no-cpu --steps 300 --warmup 30 --size $3 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print('$1', j['ms_per_step'], 'ms/step; kernel avg', r['kernel_ms_avg'], 'min', r['kernel_ms_min'], 'frac', r['frac'])"; }
t "bc1 4096" bc1 4096; t "bc3 4096" bc3 4096; t "bc1 4096" bc1 4096; t "bc3 4096" bc3 4096
