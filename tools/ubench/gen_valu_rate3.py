import re
"""Generates valu_rate.hip: issue-cost microbenchmark of the VALU instructions the BCn kernels are made of (gfx950).
Each kernel runs LOOPS x 64 copies of one instruction over 8 independent destination registers (no dependent
chain shorter than 8 instructions); one 256-thread workgroup per CU x 8, so every SIMD holds 2..8 waves of it.
cycles/instr/SIMD = elapsed_cycles * n_simd / total wave-instructions.
"""
OPS = {
    "cndmask_e32_vcc":      "v_cndmask_b32 {d}, {a}, {b}, vcc",
    "cndmask_e64_vcc":      "v_cndmask_b32_e64 {d}, {a}, {b}, vcc",
    "cndmask_e64_sgpr":     "v_cndmask_b32_e64 {d}, {a}, {b}, s[6:7]",
    "cmp+cndmask_vcc (2 inst)":  "v_cmp_lt_f32 vcc, {a}, {b}\\n\\tv_cndmask_b32 {d}, {a}, {b}, vcc",
    "cmp+nop1+cndmask_vcc":  "v_cmp_lt_f32 vcc, {a}, {b}\\n\\ts_nop 1\\n\\tv_cndmask_b32 {d}, {a}, {b}, vcc",
    "cmp64+cndmask_sgpr (2 inst)": "v_cmp_lt_f32_e64 s[6:7], {a}, {b}\\n\\ts_nop 1\\n\\tv_cndmask_b32_e64 {d}, {a}, {b}, s[6:7]",
    "cndmask_e32+add (2 inst)":   "v_cndmask_b32 {d}, {a}, {b}, vcc\\n\\tv_add_f32 {d}, {a}, {b}",
    "add+s_add (2 inst)":    "v_add_f32 {d}, {a}, {b}\\n\\ts_add_u32 s8, s8, 1",
    "add+s_nop0 (2 inst)":   "v_add_f32 {d}, {a}, {b}\\n\\ts_nop 0",
    "add+s_nop2 (2 inst)":   "v_add_f32 {d}, {a}, {b}\\n\\ts_nop 2",
    "dot2c+s_nop2 (2 inst)": "v_dot2c_i32_i16 {d}, {a}, {b}\\n\\ts_nop 2",
    "mul+add dep (2 inst)":  "v_mul_f32 {d}, {a}, {b}\\n\\tv_add_f32 {d}, {d}, {b}",
    "dot2c dep cvt (2 inst)": "v_dot2c_i32_i16 {d}, {a}, {b}\\n\\ts_nop 2\\n\\tv_cvt_f32_i32 {d}, {d}",
    "v_min3_f32":           "v_min3_f32 {d}, {a}, {b}, {d}",
    "v_med3_f32":           "v_med3_f32 {d}, {a}, {b}, {d}",
    "v_add_f32 v,v,d":      "v_add_f32 {d}, {a}, {d}",
    "v_lshlrev_b32 1":      "v_lshlrev_b32 {d}, 1, {a}",
    "v_lshlrev_b16":        "v_lshlrev_b16 {d}, 3, {a}",
    "v_sub_u16":            "v_sub_u16 {d}, {a}, {b}",
    "v_max_i16":            "v_max_i16 {d}, {a}, {b}",
    "v_max_f16":            "v_max_f16 {d}, {a}, {b}",
    "v_add_f16":            "v_add_f16 {d}, {a}, {b}",
    "v_and_b32 lit":        "v_and_b32 {d}, 0x0f0f0f0f, {a}",
    "v_and_b32 sgpr":       "v_and_b32 {d}, s4, {a}",
    "v_add_u32 sgpr":       "v_add_u32 {d}, s4, {a}",
    "v_add_f32 sgpr":       "v_add_f32 {d}, s4, {a}",
    "v_cvt_f16_f32":        "v_cvt_f16_f32 {d}, {a}",
    "v_cvt_f32_f16":        "v_cvt_f32_f16 {d}, {a}",
}
LOOPS = 2000
out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstring>', '#include <vector>', '#include <string>',
       f'#define LOOPS {LOOPS}']
names = []
for name, tpl in OPS.items():
    if tpl is None:
        continue
    fn = "k_" + re.sub(r"[^A-Za-z0-9_]", "_", name)
    names.append((name, fn))
    body = []
    for i in range(64):
        r = 2 * (i % 8)
        body.append(tpl.format(d=f"v{10 + r}", a=f"v{30 + r}", b=f"v{50 + r}",
                               d2=f"v[{10 + r}:{11 + r}]", a2=f"v[{30 + r}:{31 + r}]", b2=f"v[{50 + r}:{51 + r}]"))
    asm = "\\n\\t".join(body)
    clob = ", ".join(f'"v{i}"' for i in list(range(10, 26)) + list(range(30, 46)) + list(range(50, 66))) + ', "vcc", "scc", "s4", "s5", "s6", "s7", "s8", "s9"'
    out.append(f'''__global__ void __launch_bounds__(256) {fn}(float* o, int n) {{
    for (int i = 0; i < n; i++) asm volatile("{asm}" ::: {clob});
    if (o == nullptr) o[threadIdx.x] = 0.f;
}}''')
out.append('''int main(int argc, char** argv) {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount; const double ghz = p.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz nominal\\n", p.name, cus, ghz);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    struct K { const char* name; void (*fn)(float*, int); };
    std::vector<K> ks = {''')
for name, fn in names:
    out.append(f'        {{"{name}", {fn}}},')
out.append('''    };
    for (int wpc : {1, 2}) {                      // workgroups (of 4 waves) per CU = waves per SIMD
        printf("--- %d wave(s) per SIMD\\n", wpc);
        for (auto& k : ks) {
            float dummy; (void)dummy;
            float* d; hipMalloc(&d, 1024);
            hipLaunchKernelGGL(k.fn, dim3(cus * wpc), dim3(256), 0, 0, d, 10);
            hipDeviceSynchronize();
            hipEventRecord(a);
            hipLaunchKernelGGL(k.fn, dim3(cus * wpc), dim3(256), 0, 0, d, LOOPS);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double inst_per_simd = (double)LOOPS * 64 * wpc;       // wave-instructions issued on one SIMD
            printf("%-20s %8.3f ms  %6.2f ns/1k-inst/SIMD  => %5.2f cycles/inst @%.2f GHz\\n", k.name, ms,
                   ms * 1e6 / inst_per_simd * 1e3 / 1e3, ms * 1e-3 * ghz * 1e9 / inst_per_simd, ghz); fflush(stdout);
            hipFree(d);
        }
    }
    return 0;
}''')
open("valu_rate3.hip", "w").write("\n".join(out) + "\n")
print("wrote valu_rate.hip with", len(names), "kernels")
