cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity_bc7.py tests/test_gpu_bc7_paths.py -x -q 2>&1 | tail -3
for f in 1 0; do echo "ITW_BC7_FUSED=$f"; ITW_BC7_FUSED=$f ITW_BC7_PATH=deep timeout 600 python tools/profile_table.py 2>&1 | grep -v amdgpu | head -30; done
cd /tmp && export TMPDIR=/tmp
for prof in slow; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d /tmp/pmc -o pmc -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py 4096 $prof deep > /dev/null 2>&1
    f=$(find /tmp/pmc -name '*counter_collection*.csv' | head -1)
    python3 - "$f" $ctr $prof <<'PY'
import csv,sys
tot=0; calls=11
for r in csv.DictReader(open(sys.argv[1])):
    if 'bc7' in r['Kernel_Name'] and r['Counter_Name']==sys.argv[2]: tot+=float(r['Counter_Value'])
b=tot/calls*1024*(2 if sys.argv[2]=='FETCH_SIZE' else 1)
print(sys.argv[3], 'fused', sys.argv[2], 'per call: %.1f MB' % (b/1e6), '= %.2fx of 83.9 MB algorithmic' % (b/83.9e6))
PY
  done
done
