/*
 * extract_x86_luts.c -- derive the table model of Intel's RCPPS / RSQRTPS seed
 * approximations and prove it exhaustively against the instructions on the CPU
 * this runs on.
 *
 * Why: the reference build (IntelTextureWorks.vcxproj:388, `--opt=fast-math`)
 * turns every float divide of kernel.ispc into x*rcp(y), and ISPC's x86 rcp()/
 * rsqrt() are the hardware seed + one Newton-Raphson step.  The seeds are pure
 * table functions on Intel CPUs, so the whole arithmetic can be reproduced
 * bit-for-bit on a GPU from two 2048-entry tables.  This tool
 *   1. samples the seeds at one base exponent (rcp: 2048 mantissa buckets,
 *      rsqrt: 2 exponent parities x 1024 buckets),
 *   2. checks the closed model  seed(x) = T[bucket(x)] rescaled by exponent,
 *      incl. zero/denormal/inf/NaN/negative/underflow handling, against the
 *      instruction for ALL 2^32 float bit patterns,
 *   3. writes the tables as a C header (same file feeds oracle/ and csrc/).
 *
 * Build/run:  gcc -O2 -msse2 -fopenmp tools/extract_x86_luts.c -o /tmp/xlut &&
 *             /tmp/xlut oracle/x86_luts.h intel-texture-works-plugin_amd/csrc/x86_luts_packed.h
 * The committed headers were generated on "Intel(R) Xeon(R) Processor @ 2.10GHz"
 * (Sapphire-Rapids class).  AMD CPUs use different seeds; the tool reports a
 * mismatch count instead of silently emitting tables that do not reproduce.
 */
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <xmmintrin.h>

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline uint32_t hw_rcp(uint32_t x)   { return f2u(_mm_cvtss_f32(_mm_rcp_ss(_mm_set_ss(u2f(x))))); }
static inline uint32_t hw_rsqrt(uint32_t x) { return f2u(_mm_cvtss_f32(_mm_rsqrt_ss(_mm_set_ss(u2f(x))))); }

static uint32_t RCP_T[2048];   /* result bits for x = 1.bucket * 2^0  (exp field 127) */
static uint32_t RSQ_T[2048];   /* [0..1023]: exp field 127 (even power), [1024..2047]: exp field 128 */

static inline uint32_t model_rcp(uint32_t x)
{
    uint32_t s = x & 0x80000000u, e = (x >> 23) & 255u, m = x & 0x7fffffu;
    if (e == 255u) return m ? (x | 0x00400000u) : s;      /* NaN -> quiet NaN, inf -> signed 0 */
    if (e == 0u)   return s | 0x7f800000u;                /* zero and denormals (DAZ) -> signed inf */
    uint32_t t = RCP_T[m >> 12];                          /* exponent field of t is 126 or 127 */
    int32_t  re = (int32_t)((t >> 23) & 255u) + 127 - (int32_t)e;
    if (re <= 0) return s;                                /* result would be denormal: flushed */
    return s | ((uint32_t)re << 23) | (t & 0x7fffffu);
}

static inline uint32_t model_rsqrt(uint32_t x)
{
    uint32_t s = x & 0x80000000u, e = (x >> 23) & 255u, m = x & 0x7fffffu;
    if (e == 255u && m) return x | 0x00400000u;           /* NaN */
    if (e == 0u)        return s | 0x7f800000u;           /* +-0, +-denormal -> +-inf */
    if (s)              return 0xffc00000u;               /* negative -> default NaN */
    if (e == 255u)      return 0u;                        /* +inf -> +0 */
    uint32_t odd = (e & 1u) ? 0u : 1u;                    /* exp field 127 (odd field) = even power */
    uint32_t t = RSQ_T[(odd << 10) | (m >> 13)];
    int32_t  k = ((int32_t)e - (int32_t)(127 + odd)) / 2; /* exact: e-(127+odd) is even */
    return t - ((uint32_t)k << 23);
}

int main(int argc, char** argv)
{
    for (uint32_t i = 0; i < 2048; i++) RCP_T[i] = hw_rcp((127u << 23) | (i << 12));
    for (uint32_t i = 0; i < 1024; i++) RSQ_T[i]        = hw_rsqrt((127u << 23) | (i << 13));
    for (uint32_t i = 0; i < 1024; i++) RSQ_T[1024 + i] = hw_rsqrt((128u << 23) | (i << 13));

    unsigned long long bad_rcp = 0, bad_rsq = 0;
    #pragma omp parallel for reduction(+:bad_rcp,bad_rsq) schedule(static)
    for (long long hi = 0; hi < 65536; hi++)
        for (uint32_t lo = 0; lo < 65536; lo++) {
            uint32_t x = ((uint32_t)hi << 16) | lo;
            uint32_t a = hw_rcp(x), b = model_rcp(x);
            uint32_t c = hw_rsqrt(x), d = model_rsqrt(x);
            /* NaN payloads: compare exactly as well */
            if (a != b) { if (bad_rcp < 1) {} bad_rcp++; }
            if (c != d) bad_rsq++;
        }
    fprintf(stderr, "exhaustive 2^32 check: rcp mismatches %llu, rsqrt mismatches %llu\n", bad_rcp, bad_rsq);
    if (bad_rcp || bad_rsq) {
        /* print a few for diagnosis */
        int shown = 0;
        for (uint64_t x = 0; x < (1ull << 32) && shown < 10; x += 977) {
            uint32_t a = hw_rcp((uint32_t)x), b = model_rcp((uint32_t)x);
            if (a != b) { fprintf(stderr, " rcp   x=%08x hw=%08x model=%08x\n", (uint32_t)x, a, b); shown++; }
        }
        shown = 0;
        for (uint64_t x = 0; x < (1ull << 32) && shown < 10; x += 977) {
            uint32_t c = hw_rsqrt((uint32_t)x), d = model_rsqrt((uint32_t)x);
            if (c != d) { fprintf(stderr, " rsqrt x=%08x hw=%08x model=%08x\n", (uint32_t)x, c, d); shown++; }
        }
        return 1;
    }

    /* how many low mantissa bits are always zero (for compact storage) */
    uint32_t orr = 0, orq = 0;
    for (int i = 0; i < 2048; i++) { orr |= RCP_T[i] & 0x7fffffu; orq |= RSQ_T[i] & 0x7fffffu; }
    fprintf(stderr, "mantissa OR: rcp %06x rsqrt %06x\n", orr, orq);

    if (argc > 1) {
        FILE* f = fopen(argv[1], "w");
        if (!f) { perror("open"); return 2; }
        fprintf(f, "/* GENERATED by tools/extract_x86_luts.c -- do not edit.\n"
                   " * Intel RCPPS / RSQRTPS seed tables, verified against the instructions for all 2^32 inputs.\n"
                   " * X86_RCP_SEED[b]   = bits of rcpps(1.b * 2^0), b = top 11 mantissa bits.\n"
                   " * X86_RSQRT_SEED[p*1024+b] = bits of rsqrtps(1.b * 2^p), b = top 10 mantissa bits, p = 0/1.\n */\n");
        fprintf(f, "static const unsigned int X86_RCP_SEED[2048] = {\n");
        for (int i = 0; i < 2048; i++) fprintf(f, "0x%08xu,%s", RCP_T[i], (i % 8 == 7) ? "\n" : " ");
        fprintf(f, "};\nstatic const unsigned int X86_RSQRT_SEED[2048] = {\n");
        for (int i = 0; i < 2048; i++) fprintf(f, "0x%08xu,%s", RSQ_T[i], (i % 8 == 7) ? "\n" : " ");
        fprintf(f, "};\n");
        fclose(f);
    }
    if (argc > 2) {
        /* device form: every seed lies in [0x3f000000, 0x40000000) with the low 11
         * mantissa bits clear, so (seed - 0x3f000000) >> 11 fits 13 bits -> uint16,
         * 4 KiB per table (both tables = 8 KiB of LDS / L1). */
        for (int i = 0; i < 2048; i++) {
            uint32_t a = RCP_T[i] - 0x3f000000u, b = RSQ_T[i] - 0x3f000000u;
            if ((a & 0x7ffu) || (a >> 24) || (b & 0x7ffu) || (b >> 24)) { fprintf(stderr, "seed %d not packable\n", i); return 3; }
        }
        FILE* f = fopen(argv[2], "w");
        if (!f) { perror("open"); return 2; }
        fprintf(f, "/* GENERATED by tools/extract_x86_luts.c -- do not edit.\n"
                   " * Packed Intel RCPPS / RSQRTPS seed tables: seed_bits = 0x3f000000 | (entry << 11).\n */\n");
        fprintf(f, "X86_LUT_QUAL unsigned short X86_RCP_SEED16[2048] = {\n");
        for (int i = 0; i < 2048; i++) fprintf(f, "0x%04x,%s", (RCP_T[i] - 0x3f000000u) >> 11, (i % 16 == 15) ? "\n" : " ");
        fprintf(f, "};\nX86_LUT_QUAL unsigned short X86_RSQRT_SEED16[2048] = {\n");
        for (int i = 0; i < 2048; i++) fprintf(f, "0x%04x,%s", (RSQ_T[i] - 0x3f000000u) >> 11, (i % 16 == 15) ? "\n" : " ");
        fprintf(f, "};\n");
        fclose(f);
    }
    return 0;
}
