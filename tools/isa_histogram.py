"""Opcode histogram of a kernel's ISA, split by issue cost ("commit an ISA histogram of bc13_kernel").

Compiles the source to gfx950 assembly with the library's flags (hipcc -S, device only), cuts out every kernel whose mangled
name contains the pattern and counts its VALU opcodes in three classes, by the measured issue costs of
profiles/valu_issue_costs.md (tools/ubench): 2-cycle VOP2 forms, 4-cycle forms (everything else), 8-cycle (v_rcp / v_sqrt ...).
STATIC counts: every instruction of the kernel once, both sides of every branch, prologue included -- the executed count per
block comes from the SQ_INSTS_VALU pass (profiles/*_valu_by_workload.json) and is printed beside it when given.

    python tools/isa_histogram.py intel-texture-works-plugin_amd/csrc/bc1_bc3.hip bc13_kernel [executed_per_block ...]
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-gpu-flush-denormals-to-zero",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", "--cuda-device-only", "-S"]
TWO = {"v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32", "v_mov_b64",
       "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32", "v_cndmask_b32", "v_add_u16", "v_sub_u16",
       "v_mul_lo_u16", "v_ashrrev_i16", "v_mul_f16", "v_mul_legacy_f32"}
EIGHT = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_rcp_iflag_f32", "v_mad_u64_u32", "v_mul_hi_u32"}


def main():
    src, pat = sys.argv[1], sys.argv[2]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + FLAGS + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().split("\n")
    kernels, cur, name = {}, None, None
    for line in text:
        m = re.match(r"^(_Z\w+):", line)
        if m and pat in m.group(1):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):             # (an early `return` puts an s_endpgm in the middle of the function)
                kernels[name] = cur
                cur = None
            else:
                cur.append(line)
    demangle = lambda n: subprocess.run(["c++filt", n], stdout=subprocess.PIPE, text=True).stdout.split("(")[0].strip()
    for i, (n, body) in enumerate(kernels.items()):
        ops = collections.Counter()
        for line in body:
            m = re.match(r"\s+(v_[a-z0-9_]+)", line)
            if m:
                ops[re.sub(r"_e32$|_e64$|_sdwa$|_dpp$", "", m.group(1))] += 1
        cls = lambda o: 8 if o in EIGHT else 2 if o in TWO else 4
        tot = {c: sum(v for o, v in ops.items() if cls(o) == c) for c in (2, 4, 8)}
        n_all = sum(tot.values())
        print(f"## {demangle(n)}")
        print(f"static VALU instructions {n_all}: 2-cycle forms {tot[2]}, 4-cycle forms {tot[4]}, 8-cycle {tot[8]}"
              f"  -> {2 * tot[2] + 4 * tot[4] + 8 * tot[8]} issue cycles if every instruction ran once")
        if len(sys.argv) > 3 + i:
            print(f"executed per block (SQ_INSTS_VALU / block-waves): {sys.argv[3 + i]}")
        for c in (2, 4, 8):
            row = sorted(((v, o) for o, v in ops.items() if cls(o) == c), reverse=True)
            print(f"  {c}-cycle: " + "  ".join(f"{o} {v}" for v, o in row))
        print()


if __name__ == "__main__":
    main()
