"""tools/summarize_profiles.py normalises a profiled run's counters by the number of C-ABI calls the run MADE (`abi_calls` of the bench
line kept beside the counter file), never by the dispatch count of the least frequent kernel: since a call is two bands on two streams
every kernel runs at least twice per call and nothing runs once ("What's weak" 2: alpha_slow's VALU figure was halved)."""
import csv
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

HEADER = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name", "Workgroup_Size",
          "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def _two_band_csv(path, calls):
    """`calls` C-ABI calls of an alpha_slow-like chain: every kernel is dispatched TWICE per call (one per band), 1 ms / 0.5 ms each."""
    kernels = [("void itw::bc7_finish_all<true, 6>(unsigned char const*, long)", 400.0, 1_000_000, 243, 61440),
               ("void itw::bc7_scan_all<true, false, true>(unsigned char const*, long)", 1000.0, 500_000, 128, 36864)]
    t, disp = 10_000, 0
    with open(path, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(HEADER)
        for _ in range(calls):
            for band in range(2):
                for name, valu, ns, vgpr, lds in kernels:
                    disp += 1
                    for counter, value in (("SQ_INSTS_VALU", valu), ("SQ_WAVES", 8.0)):
                        w.writerow([disp, disp, "Agent 2", 1, 1, 1, 1024, 1, name, 256, lds, 0, vgpr, 0, 100, counter, value, t, t + ns])
                    t += ns + 1000


def test_counters_are_divided_by_the_calls_the_run_made(tmp_path):
    import summarize_profiles as sp
    src, dst = tmp_path / "prof", tmp_path / "profiles"
    src.mkdir()
    (src / "source_sha256.txt").write_text("abc\n")
    _two_band_csv(src / "pmc_sq_bc7_alpha_slow.csv", calls=7)
    (src / "pmc_sq_bc7_alpha_slow.json").write_text("some warning line\n" + json.dumps({"metric": "x", "abi_calls": 7}) + "\n")
    sp.summarize("t00", str(src), str(dst))
    j = json.load(open(dst / "t00_valu_by_workload.json"))
    row = j["bc7_alpha_slow"]
    assert row["abi_calls"] == 7 and j["_source_sha256"] == "abc"
    assert row["SQ_INSTS_VALU"] == pytest.approx(2 * (400.0 + 1000.0))            # two bands per call -- min(by_kernel) would have said 1400
    assert row["per_kernel"]["itw::bc7_finish_all<true, 6>"] == pytest.approx(800.0)
    iso = row["isolated"]["itw::bc7_scan_all<true, false, true>"]
    assert iso["dispatches"] == 2 and iso["ms"] == pytest.approx(1.0) and iso["wave_valu"] == pytest.approx(2000.0) and iso["vgprs"] == 128
    assert row["isolated"]["itw::bc7_finish_all<true, 6>"]["lds_bytes"] == 61440
    assert os.path.exists(dst / "t00_counters" / "pmc_sq_bc7_alpha_slow.csv")


def test_a_counter_file_without_its_bench_line_is_refused(tmp_path):
    import summarize_profiles as sp
    src = tmp_path / "prof"
    src.mkdir()
    _two_band_csv(src / "pmc_sq_bc7_slow.csv", calls=3)
    with pytest.raises(SystemExit):
        sp.summarize("t00", str(src), str(tmp_path / "out"))
    sp.summarize("t00", str(src), str(tmp_path / "out"), {"pmc_sq_bc7_slow": 3})      # --calls pmc_sq_bc7_slow=3
    j = json.load(open(tmp_path / "out" / "t00_valu_by_workload.json"))
    assert j["bc7_slow"]["SQ_INSTS_VALU"] == pytest.approx(2800.0)
