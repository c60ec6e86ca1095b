"""Writes tests/golden/ref_presets.bin: the 15 quality presets exactly as the REFERENCE's own translation unit
(/root/reference/3rdParty/Intel/Source/ispc_texcomp.cpp:20-410, compiled unmodified into oracle/_ref/ by
oracle/ref_build/Makefile) fills them, each over 0xA5-filled storage so that fields and padding the reference leaves
unwritten stay visible: 10 x 64 bytes (bc7_enc_settings: ultrafast, veryfast, fast, basic, slow, alpha_*) then
5 x 16 bytes (bc6h_enc_settings: veryfast, fast, basic, slow, veryslow).  A reference-held vector: unlike the block
goldens (oracle-generated), these bytes come from reference code.  Needs /root/reference (build container only)."""
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(["make", "-C", os.path.join(ROOT, "oracle")], check=True)
subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "ref_build")], check=True)
out = os.path.join(ROOT, "tests", "golden", "ref_presets.bin")
subprocess.run([os.path.join(ROOT, "oracle", "_ref", "ref_header_caller_cpu"), "profiles", out], check=True)
b = open(out, "rb").read()
assert len(b) == 10 * 64 + 5 * 16
print(out, len(b), "bytes sha256", hashlib.sha256(b).hexdigest())
