"""The REFERENCE's own dispatch code (win32Threads.cpp compiled unmodified into oracle/_ref/ref_threads_caller_gpu,
slice loop restated from IntelPlugin.cpp:851-879) driving libispc_texcomp.so with host pointers at 4096^2: the
"plugin calls the ABI unchanged" rate.  Prints one JSON object per line.  Runs on the GPU box (binaries are prebuilt by
build(); /root/reference is not needed).  Usage: python tools/ref_caller_timing.py [size] [workers,workers,...]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "intel-texture-works-plugin_amd"))
import numpy as np                      # noqa: E402
from itw_amd import surfaces           # noqa: E402

EXE = os.path.join(ROOT, "oracle", "_ref", "ref_threads_caller_gpu")
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
workers = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8, 64]
tmp = tempfile.mkdtemp()
ldr, hdr = os.path.join(tmp, "ldr.raw"), os.path.join(tmp, "hdr.raw")
surfaces.ldr_smooth(size, size).tofile(ldr)
surfaces.hdr_smooth(size, size).tofile(hdr)
out = os.path.join(tmp, "out.bin")
for tramp in ("BC1", "BC3", "BC7_basic", "BC7_slow", "BC7_alpha_basic", "BC6H_slow", "BC6H_fast"):
    src = hdr if tramp.startswith("BC6H") else ldr
    for mode, w in [("st", 1)] + [("mt", w) for w in workers if w > 1]:
        for whole in (False, True):
            best = None
            for rep in range(1):
                r = subprocess.run([EXE, mode, tramp, str(size), str(size), src, out] + (["whole"] if whole else []),
                                   capture_output=True, text=True, timeout=600, env=dict(os.environ, ITW_REF_THREADS=str(w), ITW_REF_REPS="4"))
                if r.returncode != 0:
                    print(json.dumps({"trampoline": tramp, "mode": mode, "workers": w, "error": r.stderr[-300:]}), flush=True)
                    break
                j = json.loads(r.stdout.strip().splitlines()[-1])
                if best is None or j["ms"] < best["ms"]:
                    best = j
            if best:
                best["whole_surface_call"] = whole
                print(json.dumps(best), flush=True)
