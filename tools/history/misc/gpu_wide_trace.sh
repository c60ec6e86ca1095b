#!/bin/bash
# kernel timeline of one 16 384-block wide `slow` call (rocprofv3 --kernel-trace): where do the 0.18 ms go?
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/wide_trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/raw -o t -- python $GRAFT_REPO_ROOT/tools/wide_trace_probe.py 64 slow wide > /dev/null 2> $O/log.txt
f=$(find $O/raw -name '*kernel_trace*.csv' | head -1)
python3 - "$f" <<'PY' | tee $O/timeline.txt
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = [r for r in rows if "bc7_" in r["Kernel_Name"]]
# last full call: the last 4 kernels
calls = [rows[i:i + 4] for i in range(0, len(rows) - len(rows) % 4, 4)]
for c in calls[-3:]:
    t0 = min(int(r["Start_Timestamp"]) for r in c)
    print("call:")
    for r in sorted(c, key=lambda r: int(r["Start_Timestamp"])):
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        print(f"   {r['Kernel_Name'].split('(')[0][-40:]:40s} start {s / 1e3:8.1f} us  end {e / 1e3:8.1f} us  dur {(e - s) / 1e3:7.1f}  grid {r.get('Grid_Size','')}")
    print(f"   span {(max(int(r['End_Timestamp']) for r in c) - t0) / 1e3:.1f} us")
PY
rm -rf $O/raw
