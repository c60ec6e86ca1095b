cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r02e; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "64 slow bc7" "64 basic bc7" "8 slow bc7" "64 slow bc6h" "64 fast bc6h"; do
  set -- $cfg
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py $1 $2 wide $3 > /dev/null 2> $OUT/err.log
  echo "== rows=$1 $2 $3"; find $OUT/t -name '*kernel_stats*.csv' | head -1 | xargs python3 -c "
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    n=r['Name']
    if 'bc7' in n or 'bc6h' in n: print(n[:60].ljust(60), r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
"
  rm -rf $OUT/t
done
