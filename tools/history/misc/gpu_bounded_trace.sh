# per-kernel times of the bounded BC7 order (rocprofv3 kernel trace) on the bench surface and on a natural image
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/bounded
for run in "slow I3" "slow baboon" "alpha_slow I3opaque"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o tr -- python $R/tools/bc7_trace_run.py $run > /dev/null 2> /tmp/tr.log
  f=$(find /tmp/tr -name '*kernel_stats*.csv' | head -1)
  echo "== $run"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "bc7" in r["Name"]: print(f'{r["Name"][:70]:70s} calls {r["Calls"]:>3s} avg {float(r["AverageNs"])/1e6:8.3f} ms')
PY
  rm -rf /tmp/tr
done 2>&1 | tee $R/gpurun_out/bounded/kernel_times.txt
