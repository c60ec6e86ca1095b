# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of one BC7 call at 4096^2 by launch shape
cd $GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for prof in slow alpha_slow; do for path in deep wide; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc -o pmc -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py 4096 $prof $path > /dev/null 2>&1
    f=$(find /tmp/pmc -name '*counter_collection*.csv' | head -1)
    python3 - "$f" $ctr $prof $path <<'PY'
import csv,sys
tot=0; calls=11
for r in csv.DictReader(open(sys.argv[1])):
    if 'bc7' in r['Kernel_Name'] and r['Counter_Name']==sys.argv[2]: tot+=float(r['Counter_Value'])
kib=tot/calls
b=kib*1024*(2 if sys.argv[2]=='FETCH_SIZE' else 1)
print(sys.argv[3], sys.argv[4], sys.argv[2], 'per call: %.1f MB' % (b/1e6), '= %.2fx of 83.9 MB algorithmic' % (b/83.9e6))
PY
  done
done; done
cd $GRAFT_REPO_ROOT
for path in deep wide; do ITW_BC7_PATH=$path python tools/host_path_timing.py 2>&1 | grep "bc7"; done
