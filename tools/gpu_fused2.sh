cd /tmp && export TMPDIR=/tmp
for g in 0 8 256 1024 2048; do
ITW_SCAN_GRAIN=$g rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/t -o t -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py 4096 slow deep > /dev/null 2>&1
echo "== ITW_SCAN_GRAIN=$g"; find /tmp/t -name '*kernel_stats*.csv' | head -1 | xargs python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'bc7_scan' in n: print(n[:50].ljust(50), r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
"
rm -rf /tmp/t
rm -rf /tmp/pmc
ITW_SCAN_GRAIN=$g rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmc -o pmc -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py 4096 slow deep > /dev/null 2>&1
f=$(find /tmp/pmc -name '*counter_collection*.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys,collections
tot=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'bc7' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE': tot[r['Kernel_Name'][10:30]]+=float(r['Counter_Value'])
for k,v in tot.items(): print('   FETCH', k, '%.1f MB per call' % (v/11*1024*2/1e6))
PY
done
