"""Generates valu_rate.hip: issue-cost microbenchmark of the VALU instructions the BCn kernels are made of (gfx950).
Each kernel runs LOOPS x 64 copies of one instruction over 8 independent destination registers (no dependent
chain shorter than 8 instructions); one 256-thread workgroup per CU x 8, so every SIMD holds 2..8 waves of it.
cycles/instr/SIMD = elapsed_cycles * n_simd / total wave-instructions.
"""
OPS = {
    "v_mul_f32":        "v_mul_f32 {d}, {a}, {b}",
    "v_fma_f32":        "v_fma_f32 {d}, {a}, {b}, {d}",
    "v_sub_f32":        "v_sub_f32 {d}, {a}, {b}",
    "v_subrev_f32":     "v_subrev_f32 {d}, {a}, {b}",
    "v_fmac_f32":       "v_fmac_f32 {d}, {a}, {b}",
    "v_mul_f32_e64":    "v_mul_f32_e64 {d}, {a}, {b}",
    "v_mul_f32 neg":    "v_mul_f32_e64 {d}, -{a}, {b}",
    "v_mul_f32 lit":    "v_mul_f32 {d}, 0x3b808081, {b}",
    "v_add_f32 inl":    "v_add_f32 {d}, 0.5, {b}",
    "v_mul_f32 sgpr":   "v_mul_f32 {d}, s4, {b}",
    "v_min_f32":        "v_min_f32 {d}, {a}, {b}",
    "v_xor_b32":        "v_xor_b32 {d}, {a}, {b}",
    "v_not_b32":        "v_not_b32 {d}, {a}",
    "v_lshlrev_b32":    "v_lshlrev_b32 {d}, 3, {a}",
    "v_ashrrev_i32":    "v_ashrrev_i32 {d}, 6, {a}",
    "v_mul_u32_u24":    "v_mul_u32_u24 {d}, {a}, {b}",
    "v_min_u32":        "v_min_u32 {d}, {a}, {b}",
    "v_max_i32":        "v_max_i32 {d}, {a}, {b}",
    "v_cndmask_e32":    "v_cndmask_b32 {d}, {a}, {b}, vcc",
    "v_cndmask_e64":    "v_cndmask_b32_e64 {d}, {a}, {b}, s[6:7]",
    "v_cvt_f32_ubyte0": "v_cvt_f32_ubyte0 {d}, {a}",
    "v_cvt_u32_f32":    "v_cvt_u32_f32 {d}, {a}",
    "v_cvt_f32_u32":    "v_cvt_f32_u32 {d}, {a}",
    "v_floor_f32":      "v_floor_f32 {d}, {a}",
    "v_rndne_f32":      "v_rndne_f32 {d}, {a}",
    "v_trunc_f32":      "v_trunc_f32 {d}, {a}",
    "v_add_co_u32":     "v_add_co_u32 {d}, vcc, {a}, {b}",
    "v_addc_co_u32":    "v_addc_co_u32 {d}, vcc, {a}, {b}, vcc",
    "v_bfi_b32":        "v_bfi_b32 {d}, {a}, {b}, {d}",
    "v_and_or_b32":     "v_and_or_b32 {d}, {a}, {b}, {d}",
    "v_or3_b32":        "v_or3_b32 {d}, {a}, {b}, {d}",
    "v_add3_u32":       "v_add3_u32 {d}, {a}, {b}, {d}",
    "v_add_lshl_u32":   "v_add_lshl_u32 {d}, {a}, {b}, 2",
    "v_alignbit_b32":   "v_alignbit_b32 {d}, {a}, {b}, 8",
    "v_cvt_pk_u8_f32":  "v_cvt_pk_u8_f32 {d}, {a}, 1, {d}",
    "v_pk_mul_lo_u16":  "v_pk_mul_lo_u16 {d}, {a}, {b}",
    "v_pk_min_i16":     "v_pk_min_i16 {d}, {a}, {b}",
    "v_add_u16":        "v_add_u16 {d}, {a}, {b}",
    "v_mul_lo_u16":     "v_mul_lo_u16 {d}, {a}, {b}",
    "v_mad_u16":        "v_mad_u16 {d}, {a}, {b}, {d}",
    "v_ashrrev_i16":    "v_ashrrev_i16 {d}, 6, {a}",
    "v_mul_f16":        "v_mul_f16 {d}, {a}, {b}",
    "v_add_u32_sdwa":   "v_add_u32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD",
    "v_sub_f32_sdwa":   "v_sub_f32_sdwa {d}, {a}, {b} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD",
    "v_cvt_f32_u32_sdwa": "v_cvt_f32_u32_sdwa {d}, {a} dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2",
    "v_mov_b32_sdwa":   "v_mov_b32_sdwa {d}, {a} dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1",
    "v_mov_b32_dpp":    "v_mov_b32_dpp {d}, {a} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
    "v_mul_legacy_f32": "v_mul_legacy_f32 {d}, {a}, {b}",
    "v_mad_legacy?":    None,
    "v_ldexp_f32":      "v_ldexp_f32 {d}, {a}, {b}",
    "v_max3_f32":       "v_max3_f32 {d}, {a}, {b}, {d}",
    "v_sub_u32 lit":    "v_sub_u32 {d}, 0x01010101, {b}",
    "v_subrev_u32":     "v_subrev_u32 {d}, {a}, {b}",
    "v_mad_f32?":       None,
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
    for (int wpc : {2}) {                      // workgroups (of 4 waves) per CU = waves per SIMD
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
open("valu_rate2.hip", "w").write("\n".join(out) + "\n")
print("wrote valu_rate.hip with", len(names), "kernels")
