cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t -o t -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py 4096 slow deep > /dev/null 2>&1
find /tmp/t -name '*kernel_stats*.csv' | head -1 | xargs python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'bc7_' in n: print(n[:50].ljust(50), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
"
rm -rf /tmp/t
for ctr in FETCH_SIZE WRITE_SIZE; do
rm -rf /tmp/pmc
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc -o pmc -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py 4096 slow deep > /dev/null 2>&1
f=$(find /tmp/pmc -name '*counter_collection*.csv' | head -1)
python3 - "$f" $ctr <<'PY'
import csv,sys,collections
tot=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'bc7' in r['Kernel_Name'] and r['Counter_Name']==sys.argv[2]: tot[r['Kernel_Name'][10:30]]+=float(r['Counter_Value'])
for k,v in tot.items(): print('   ', sys.argv[2], k, '%.1f MB per call' % (v/11*1024*(2 if sys.argv[2]=='FETCH_SIZE' else 1)/1e6))
PY
done
cd $GRAFT_REPO_ROOT; timeout 300 python tools/bc7_path_probe.py slow,basic 2>&1 | grep -E " 64 | 4096 "
