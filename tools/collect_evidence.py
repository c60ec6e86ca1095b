"""Copies what `tools/evidence.sh <tag>` left under gpurun_out/evidence_<tag>/ into profiles/<tag>_<name> (the rocprofv3 summaries are
tools/summarize_profiles.py's job).  Usage: python tools/collect_evidence.py <tag>"""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
src = os.path.join(ROOT, "gpurun_out", "evidence_" + tag)
dst = os.path.join(ROOT, "profiles")
for name in sorted(os.listdir(src)):
    p = os.path.join(src, name)
    if os.path.isfile(p) and os.path.getsize(p) > 0 and not name.endswith(".err"):
        shutil.copy(p, os.path.join(dst, f"{tag}_{name}"))
        print("profiles/" + f"{tag}_{name}")
