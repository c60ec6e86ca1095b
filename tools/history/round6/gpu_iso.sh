#!/bin/bash
# isolated (serialised) per-kernel durations + VALU counts of the given workloads: tools/history/round6/gpu_iso.sh <tag> wl1 wl2 ...
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in "$@"; do
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY -d $OUT/pmc_sq -o pmc -- python $ROOT/bench.py --workload $wl --no-formats --no-cpu --steps 2 --warmup 1 > $OUT/pmc_sq_$wl.json 2> $OUT/pmc_sq_$wl.log
  f=$(find $OUT/pmc_sq -name '*counter_collection*.csv' | head -1)
  if [ -n "$f" ]; then head -1 $f > $OUT/pmc_sq_$wl.csv; grep -E 'bc7_|bc6h_|bc13' $f | head -8000 >> $OUT/pmc_sq_$wl.csv; fi
  rm -rf $OUT/pmc_sq
  python3 - $OUT/pmc_sq_$wl.csv $OUT/pmc_sq_$wl.json $wl <<'PY'
import csv, json, sys, collections
calls = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])["abi_calls"]
seen = {}
for r in csv.DictReader(open(sys.argv[1])):
    d = seen.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"].replace("void itw::", "").split("(")[0], "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), "v": r["VGPR_Count"], "lds": r["LDS_Block_Size"], "c": {}})
    d["c"][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.OrderedDict()
for d in seen.values():
    a = agg.setdefault(d["k"], [0, 0.0, 0.0, d["v"], d["lds"]])
    a[0] += 1; a[1] += d["ns"] / 1e6; a[2] += d["c"].get("SQ_INSTS_VALU", 0)
print("==", sys.argv[3], "calls", calls)
tot = 0
for k, a in agg.items():
    ms = a[1] / calls; v = a[2] / calls; tot += ms
    print(f"  {k:38s} disp/call {a[0]/calls:4.1f} ms/call {ms:7.3f} waveVALU {v/1e6:8.1f}M lane-frac {v*64/(ms*1e-3)/78.6e12 if ms else 0:.3f} vgpr {a[3]} lds {a[4]}")
print(f"  sum {tot:.3f} ms")
PY
done
