"""MEASUREMENT VARIANT (never in the product build): how often could a scan stop evaluating a shape early?

VERDICT r03 item 1d: "wave-level early-out of the error accumulation: per-texel errors are non-negative and the winner test is
a strict `<` -- once every block of the wave is already above its best, the rest of the shape cannot matter; report the prune rate
on I3 / I2 first; implement only if >= 10 %".  This script writes a patched copy of csrc/bc7.hip that COUNTS, per wave and shape,
whether the exact error of the subsets evaluated so far already exceeds the incumbent for all 64 blocks (modes 1/3: after the
first of two subsets; modes 0/2: after two of three), builds gpurun_variants/lib_pruneprobe.so with it, and
tools/variants/bc7_prune_probe_run.py prints the rates on the GPU box.  The emitted blocks are unchanged."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "intel-texture-works-plugin_amd", "csrc")
src = open(os.path.join(CS, "bc7.hip")).read()


def patch(s, old, new):
    assert s.count(old) == 1, (s.count(old), old[:60])
    return s.replace(old, new)


src = patch(src, "constexpr int32_t ERR_MAX = 0x7fffffff;",
            "constexpr int32_t ERR_MAX = 0x7fffffff;\n__device__ unsigned int g_probe[8];   // [0] wave-shapes 13, [1] all-dead 13, [2] dead lanes 13, [3..5] the same for 02\n")
# modes 1/3 (table-order scan): after subset 0
src = patch(src, """                        subset_error2_pal<3, 2, 3, TPB>(ea, ec, ln.tx, s1, ln.pal, s3, ln.pal + 8 * TPB, sm.bits);
                    } else if (na > 0) {""",
            """                        subset_error2_pal<3, 2, 3, TPB>(ea, ec, ln.tx, s1, ln.pal, s3, ln.pal + 8 * TPB, sm.bits);
                        if (j == 0) {
                            const int32_t tt0 = st.m[0] + st.m[4] + st.m[7];
                            const bool dead = (ea + tt0 > wa.err) && (ec + tt0 > wb.err);
                            const unsigned long long m = __ballot(dead);
                            if ((threadIdx.x & 63u) == 0u) { atomicAdd(&g_probe[0], 1u); if (m == ~0ull) atomicAdd(&g_probe[1], 1u); atomicAdd(&g_probe[2], (unsigned)__popcll(m)); }
                        }
                    } else if (na > 0) {""")
# modes 0/2: after subset 1 (two of three), only where the remainder is known
src = patch(src, """                if (act == 1u) ITW_CACHE_PUT(ce, slot, r);
            }
        }""",
            """                if (act == 1u) ITW_CACHE_PUT(ce, slot, r);
            }
            if (j == 1 && rest_valid) {
                const int32_t tt01 = tt - (rest.m[0] + rest.m[4] + rest.m[7]);
                const bool dead = (!do0 || e0 + tt01 > b0.err) && (!do2 || e2 + tt01 > b2.err);
                const unsigned long long m = __ballot(dead);
                if ((threadIdx.x & 63u) == 0u) { atomicAdd(&g_probe[3], 1u); if (m == ~0ull) atomicAdd(&g_probe[4], 1u); atomicAdd(&g_probe[5], (unsigned)__popcll(m)); }
            }
        }""")
src += """
extern "C" void itwProbeReadCounters(unsigned int* out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(itw::g_probe), 8 * sizeof(unsigned int));
    unsigned int zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(itw::g_probe), zero, sizeof zero);
}
"""
os.makedirs("/tmp/var/pruneprobe", exist_ok=True)
for name in os.listdir(CS):
    if name.endswith((".hpp", ".h")):
        t = open(os.path.join(CS, name)).read().replace('"../../include/', '"%s/include/' % ROOT)
        open(os.path.join("/tmp/var/pruneprobe", name), "w").write(t)
open("/tmp/var/pruneprobe/bc7.hip", "w").write(src.replace('"../../include/', '"%s/include/' % ROOT))
FL = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-flush-denormals-to-zero "
      "-fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-function").split()
subprocess.run(["/opt/rocm/bin/hipcc"] + FL + ["-c", "/tmp/var/pruneprobe/bc7.hip", "-o", "/tmp/var/pruneprobe/bc7.o"], check=True)
others = [os.path.join(CS, "build", o) for o in os.listdir(os.path.join(CS, "build")) if o.endswith(".o") and o != "bc7.o"]
os.makedirs(os.path.join(ROOT, "gpurun_variants"), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "gpurun_variants", "lib_pruneprobe.so"),
                "/tmp/var/pruneprobe/bc7.o"] + others, check=True)
print("built gpurun_variants/lib_pruneprobe.so")
