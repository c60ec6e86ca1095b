"""MEASUREMENT VARIANT (never in the product build): the shader clock a kernel actually runs at.

roofline.issue prices issue cycles against 1 024 SIMDs x 2.4 GHz nominal; under load the chip may clock lower.  This script writes
patched copies of csrc/bc1_bc3.hip and csrc/bc7.hip in which thread 0 of every workgroup reads clock64() (s_memtime: shader cycles)
and wall_clock64() (100 MHz constant clock) at kernel entry and exit and adds both differences to a global pair, builds
gpurun_variants/lib_clockprobe.so, and tools/variants/clock_probe_run.py prints cycles / ns per kernel on the GPU box."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CS = os.path.join(ROOT, "intel-texture-works-plugin_amd", "csrc")
os.makedirs("/tmp/var/clockprobe", exist_ok=True)
for name in os.listdir(CS):
    if name.endswith((".hpp", ".h")):
        open(os.path.join("/tmp/var/clockprobe", name), "w").write(open(os.path.join(CS, name)).read().replace('"../../include/', '"%s/include/' % ROOT))


def patch(s, old, new, count=1):
    assert s.count(old) == count, (s.count(old), old[:70])
    return s.replace(old, new)


HEAD = "\n__device__ unsigned long long g_clk[4];   // [0] shader cycles, [1] 100 MHz ticks, summed over workgroups; [2] workgroups\n"
ENTER = "    unsigned long long clk_c0 = 0, clk_w0 = 0; if (threadIdx.x == 0) { clk_c0 = clock64(); clk_w0 = wall_clock64(); }\n"
LEAVE = "    if (threadIdx.x == 0) { atomicAdd(&g_clk[0], clock64() - clk_c0); atomicAdd(&g_clk[1], wall_clock64() - clk_w0); atomicAdd(&g_clk[2], 1ull); }\n"
READ = """
extern "C" void itwProbeReadClock%s(unsigned long long* out)
{
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(itw::g_clk), 4 * sizeof(unsigned long long));
    unsigned long long zero[4] = {0, 0, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(itw::g_clk), zero, sizeof zero);
}
"""
# BC1 / BC3
s = open(os.path.join(CS, "bc1_bc3.hip")).read().replace('"../../include/', '"%s/include/' % ROOT)
s = patch(s, "namespace itw {\n", "namespace itw {\n" + HEAD)
s = patch(s, "    __shared__ __attribute__((aligned(16))) unsigned char s_tables[BC1_LDS_BYTES];\n",
          "    __shared__ __attribute__((aligned(16))) unsigned char s_tables[BC1_LDS_BYTES];\n" + ENTER)
s = patch(s, "        if (base >= nblocks) break;                                    // wave-uniform\n",
          "        if (base >= nblocks) break;                                    // wave-uniform\n")
s = patch(s, "        load_words<VEC16>(w, src, stride, blocks_x, nxt < nblocks ? nxt : nblocks - 1);\n    }\n}\n",
          "        load_words<VEC16>(w, src, stride, blocks_x, nxt < nblocks ? nxt : nblocks - 1);\n    }\n" + LEAVE + "}\n")
s += READ % "Bc13"
open("/tmp/var/clockprobe/bc1_bc3.hip", "w").write(s)
# BC7 scan (the fused path's scan kernel): entry after the early returns, exit at the end
s = open(os.path.join(CS, "bc7.hip")).read().replace('"../../include/', '"%s/include/' % ROOT)
s = patch(s, "constexpr int32_t ERR_MAX = 0x7fffffff;", "constexpr int32_t ERR_MAX = 0x7fffffff;" + HEAD)
s = patch(s, "    if (chunk * TPB >= nact) return;\n    Lane ln;\n", "    if (chunk * TPB >= nact) return;\n" + ENTER + "    Lane ln;\n")
s = patch(s, "        if (live) wins4[(int64_t)wide_win_slot(7) * nblocks + b] = pack_win(wa);\n    }\n}\n",
          "        if (live) wins4[(int64_t)wide_win_slot(7) * nblocks + b] = pack_win(wa);\n    }\n" + LEAVE + "}\n")
s += READ % "Bc7"
open("/tmp/var/clockprobe/bc7.hip", "w").write(s)
FL = ("--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-flush-denormals-to-zero "
      "-fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize -Wno-unused-function").split()
objs = []
for f in ("bc1_bc3", "bc7"):
    subprocess.run(["/opt/rocm/bin/hipcc"] + FL + ["-c", f"/tmp/var/clockprobe/{f}.hip", "-o", f"/tmp/var/clockprobe/{f}.o"], check=True)
    objs.append(f"/tmp/var/clockprobe/{f}.o")
others = [os.path.join(CS, "build", o) for o in os.listdir(os.path.join(CS, "build")) if o.endswith(".o") and o not in ("bc7.o", "bc1_bc3.o")]
os.makedirs(os.path.join(ROOT, "gpurun_variants"), exist_ok=True)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ROOT, "gpurun_variants", "lib_clockprobe.so")] + objs + others, check=True)
print("built gpurun_variants/lib_clockprobe.so")
