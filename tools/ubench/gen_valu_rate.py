"""Generates valu_rate.hip: issue-cost microbenchmark of the VALU instructions the BCn kernels are made of (gfx950).
Each kernel runs LOOPS x 64 copies of one instruction over 8 independent destination registers (no dependent
chain shorter than 8 instructions); one 256-thread workgroup per CU x 8, so every SIMD holds 2..8 waves of it.
cycles/instr/SIMD = elapsed_cycles * n_simd / total wave-instructions.
"""
OPS = {
    "v_fma_f32":        "v_fma_f32 {d}, {a}, {b}, {d}",
    "v_mul_f32":        "v_mul_f32 {d}, {a}, {b}",
    "v_add_f32":        "v_add_f32 {d}, {a}, {b}",
    "v_pk_mul_f32":     "v_pk_mul_f32 {d2}, {a2}, {b2}",
    "v_pk_add_f32":     "v_pk_add_f32 {d2}, {a2}, {b2}",
    "v_pk_fma_f32":     "v_pk_fma_f32 {d2}, {a2}, {b2}, {d2}",
    "v_dot2c_i32_i16":  "v_dot2c_i32_i16 {d}, {a}, {b}",
    "v_dot2_i32_i16":   "v_dot2_i32_i16 {d}, {a}, {b}, {d}",
    "v_dot4_u32_u8":    "v_dot4_u32_u8 {d}, {a}, {b}, {d}",
    "v_pk_mad_u16":     "v_pk_mad_u16 {d}, {a}, {b}, {d}",
    "v_pk_ashrrev_i16": "v_pk_ashrrev_i16 {d}, 6, {a}",
    "v_pk_add_u16":     "v_pk_add_u16 {d}, {a}, {b}",
    "v_pk_sub_i16":     "v_pk_sub_i16 {d}, {a}, {b}",
    "v_perm_b32":       "v_perm_b32 {d}, {a}, {b}, {d}",
    "v_cndmask_b32":    "v_cndmask_b32 {d}, {a}, {b}, vcc",
    "v_cvt_f32_i32":    "v_cvt_f32_i32 {d}, {a}",
    "v_cvt_i32_f32":    "v_cvt_i32_f32 {d}, {a}",
    "v_cvt_f32_ubyte1": "v_cvt_f32_ubyte1 {d}, {a}",
    "v_med3_i32":       "v_med3_i32 {d}, {a}, 1, 7",
    "v_mad_u32_u24":    "v_mad_u32_u24 {d}, {a}, {b}, {d}",
    "v_mad_i32_i24":    "v_mad_i32_i24 {d}, {a}, {b}, {d}",
    "v_mul_i32_i24":    "v_mul_i32_i24 {d}, {a}, {b}",
    "v_mul_lo_u32":     "v_mul_lo_u32 {d}, {a}, {b}",
    "v_add_u32":        "v_add_u32 {d}, {a}, {b}",
    "v_sub_u32":        "v_sub_u32 {d}, {a}, {b}",
    "v_lshl_or_b32":    "v_lshl_or_b32 {d}, {a}, 8, {d}",
    "v_lshl_add_u32":   "v_lshl_add_u32 {d}, {a}, 3, {d}",
    "v_and_b32":        "v_and_b32 {d}, {a}, {b}",
    "v_or_b32":         "v_or_b32 {d}, {a}, {b}",
    "v_lshrrev_b32":    "v_lshrrev_b32 {d}, 4, {a}",
    "v_bfe_u32":        "v_bfe_u32 {d}, {a}, 8, 8",
    "v_min_i32":        "v_min_i32 {d}, {a}, {b}",
    "v_max_f32":        "v_max_f32 {d}, {a}, {b}",
    "v_cmp_lt_f32":     "v_cmp_lt_f32 vcc, {a}, {b}",
    "v_cmp_lt_i32":     "v_cmp_lt_i32 vcc, {a}, {b}",
    "v_mov_b32":        "v_mov_b32 {d}, {a}",
    "v_rcp_f32":        "v_rcp_f32 {d}, {a}",
    "v_sqrt_f32":       "v_sqrt_f32 {d}, {a}",
    "v_sad_u8":         "v_sad_u8 {d}, {a}, {b}, {d}",
    "v_mad_f32? (mul+add)": None,
}
LOOPS = 2000
out = ['#include <hip/hip_runtime.h>', '#include <cstdio>', '#include <cstring>', '#include <vector>', '#include <string>',
       f'#define LOOPS {LOOPS}']
names = []
for name, tpl in OPS.items():
    if tpl is None:
        continue
    fn = "k_" + name.replace(".", "_").replace(" ", "_").replace("?", "")
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
    for (int wpc : {1, 2, 4}) {                      // workgroups (of 4 waves) per CU = waves per SIMD
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
open("valu_rate.hip", "w").write("\n".join(out) + "\n")
print("wrote valu_rate.hip with", len(names), "kernels")
