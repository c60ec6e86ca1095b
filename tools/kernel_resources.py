"""Per-kernel register / scratch / LDS usage of a compiled .o (hipcc fat object): unbundles the gfx950 code object and reads its metadata notes.
usage: python tools/kernel_resources.py intel-texture-works-plugin_amd/csrc/build/bc7.o [name filter]"""
import re, subprocess, sys, tempfile, os
LLVM = "/opt/rocm/lib/llvm/bin"
obj = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.TemporaryDirectory() as d:
    # the device code object sits in the .hip_fatbin section of the host object
    fb = os.path.join(d, "fb"); co = os.path.join(d, "co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", obj, fb], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}", f"--output={co}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950"], check=True)
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
for k in re.split(r"\n  - (?=\.agpr_count:)", notes)[1:]:
    g = lambda key: (re.search(rf"\.{key}:\s*(\S+)", k) or [None, "?"])[1]
    name = g("name")
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    short = re.sub(r"\(.*", "", dem)
    if flt and flt not in short: continue
    print(f"{short:70s} vgpr {g('vgpr_count'):>4s} agpr {g('agpr_count'):>3s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} B  lds {g('group_segment_fixed_size'):>6s} B  spills v{g('vgpr_spill_count')} s{g('sgpr_spill_count')}")
