"""Timeline of the last BC7 call in a rocprofv3 --kernel-trace CSV: every kernel's queue, grid, start and end relative to the call's first
kernel.  usage: python tools/trace_timeline.py <kernel_trace.csv> [anchor kernel substring, default bc7_pilot_decide]"""
import csv, re, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "bc7" in r["Kernel_Name"]]
anchor = sys.argv[2] if len(sys.argv) > 2 else "bc7_pilot_decide"
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
if not idx:
    sys.exit("anchor kernel not found")
a = rows[idx[-1]]["s"]
call = [r for r in rows if a - 2_000_000 <= r["s"] <= a + 9_000_000]
# cut at gaps of more than 1.5 ms without any kernel running (call boundary) around the anchor
call.sort(key=lambda r: r["s"])
t0 = call[0]["s"]
for r in call:
    name = re.sub(r"^void itw::", "", r["Kernel_Name"]); name = re.sub(r"\(.*", "", name)
    print(f'q{r["Queue_Id"]:>2s} {name:34s} grid {int(r["Grid_Size_X"]) // 256:6d} wg  {(r["s"] - t0) / 1e6:8.3f} -> {(r["e"] - t0) / 1e6:8.3f} ms  ({(r["e"] - r["s"]) / 1e6:6.3f})')
