"""Oracle thread-scaling probe on the GPU box's host (decides how many threads the cpu_baseline leg should use)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
from oracle import pyoracle
from itw_amd import surfaces
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max:", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e:
    print("no cpu.max", e)
img = surfaces.ldr_smooth(1024, 1024)
for n in (1, 8, 32, 64, 128, 256):
    rows = min(1024, max(16, 4 * n))
    t0 = time.perf_counter(); pyoracle.encode_mt("bc7", img[:rows], "slow", threads=n); dt = time.perf_counter() - t0
    print(f"threads {n:4d} rows {rows:5d}: {dt:.3f} s  {rows*1024/dt/1e6:.3f} Mpix/s  ({dt/(rows/4*256)*1e6*n:.0f} thread-us/block)")
