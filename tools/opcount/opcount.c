/*
 * tools/opcount/opcount.c -- TEST / MEASUREMENT INFRASTRUCTURE (uses the oracle; never part of the product).
 *
 * Dynamic instruction counts of the CPU oracle, measured, not estimated ("instrument the oracle with op
 * counters and replace the estimates"): the encode call is run under ptrace single-stepping and every retired
 * instruction inside liboracle_bcn.so is histogrammed by address.  tools/opcount.py joins the histogram with
 * `objdump -d` to classify mnemonics (mulss / addss / subss / divss / sqrtss / comiss / cvttss2si ...), which gives the
 * fp32 multiplies, adds, compares and conversions the reference's algorithm executes per block as compiled from the
 * line-by-line restatement (gcc -O2, scalar SSE, no FMA, no auto-vectorisation: one x86 fp instruction = one fp32 op).
 *
 *   opcount <liboracle.so> <fmt: bc1|bc3|bc7|bc6h> <profile|-> <width> <height> <texels.raw> <histogram.out>
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/ptrace.h>
#include <sys/types.h>
#include <sys/user.h>
#include <sys/wait.h>
#include <unistd.h>

typedef struct { uint8_t* ptr; int32_t width, height, stride; } surface_t;

static int child_main(char** argv)
{
    void* L = dlopen(argv[1], RTLD_NOW);
    if (!L) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 3; }
    const char* fmt = argv[2];
    const char* prof = argv[3];
    int w = atoi(argv[4]), h = atoi(argv[5]);
    int bpp = !strcmp(fmt, "bc6h") ? 8 : 4;
    size_t n = (size_t)w * h * bpp;
    uint8_t* tex = malloc(n);
    FILE* f = fopen(argv[6], "rb");
    if (!f || fread(tex, 1, n, f) != n) { fprintf(stderr, "cannot read %s\n", argv[6]); return 3; }
    fclose(f);
    uint8_t* out = calloc((size_t)(w / 4) * (h / 4), 16);
    surface_t s = {tex, w, h, w * bpp};
    uint8_t settings[64];
    memset(settings, 0, sizeof settings);
    void (*enc2)(const surface_t*, uint8_t*) = NULL;
    void (*enc3)(const surface_t*, uint8_t*, const void*) = NULL;
    if (!strcmp(fmt, "bc1")) enc2 = dlsym(L, "oracle_CompressBlocksBC1");
    else if (!strcmp(fmt, "bc3")) enc2 = dlsym(L, "oracle_CompressBlocksBC3");
    else if (!strcmp(fmt, "bc7")) {
        int (*gp)(const char*, void*) = dlsym(L, "oracle_GetProfile_bc7");
        if (!gp || gp(prof, settings)) { fprintf(stderr, "bad bc7 profile\n"); return 3; }
        enc3 = dlsym(L, "oracle_CompressBlocksBC7");
    } else {
        int (*gp)(const char*, void*) = dlsym(L, "oracle_GetProfile_bc6h");
        if (!gp || gp(prof, settings)) { fprintf(stderr, "bad bc6h profile\n"); return 3; }
        enc3 = dlsym(L, "oracle_CompressBlocksBC6H");
    }
    if (!enc2 && !enc3) { fprintf(stderr, "symbol missing\n"); return 3; }
    raise(SIGSTOP);                                  /* marker: begin */
    if (enc2) enc2(&s, out); else enc3(&s, out, settings);
    raise(SIGSTOP);                                  /* marker: end */
    return 0;
}

/* open-addressing histogram keyed by instruction address */
#define HBITS 20
static uint64_t hk[1u << HBITS], hv[1u << HBITS];
static void bump(uint64_t rip)
{
    uint64_t i = (rip * 0x9E3779B97F4A7C15ull) >> (64 - HBITS);
    while (hk[i] && hk[i] != rip) i = (i + 1) & ((1u << HBITS) - 1);
    hk[i] = rip; hv[i]++;
}

int main(int argc, char** argv)
{
    if (argc != 8) { fprintf(stderr, "usage: %s <liboracle.so> <fmt> <profile|-> <w> <h> <texels.raw> <hist.out>\n", argv[0]); return 2; }
    pid_t pid = fork();
    if (pid == 0) {
        ptrace(PTRACE_TRACEME, 0, 0, 0);
        raise(SIGSTOP);                              /* handshake */
        _exit(child_main(argv));
    }
    int st;
    waitpid(pid, &st, 0);                            /* handshake stop */
    ptrace(PTRACE_CONT, pid, 0, 0);
    waitpid(pid, &st, 0);                            /* begin marker (or early exit) */
    if (WIFEXITED(st)) { fprintf(stderr, "child exited early (%d)\n", WEXITSTATUS(st)); return 3; }
    /* where is the oracle mapped? */
    char maps[64], line[512], want[256];
    snprintf(maps, sizeof maps, "/proc/%d/maps", (int)pid);
    const char* base_name = strrchr(argv[1], '/') ? strrchr(argv[1], '/') + 1 : argv[1];
    snprintf(want, sizeof want, "%s", base_name);
    uint64_t lo = ~0ull, hi = 0;
    FILE* m = fopen(maps, "r");
    while (m && fgets(line, sizeof line, m)) {
        if (!strstr(line, want)) continue;
        uint64_t a, b;
        if (sscanf(line, "%lx-%lx", &a, &b) == 2) { if (a < lo) lo = a; if (b > hi) hi = b; }
    }
    if (m) fclose(m);
    if (hi == 0) { fprintf(stderr, "oracle mapping not found\n"); return 3; }
    uint64_t steps = 0, inside = 0;
    for (;;) {
        if (ptrace(PTRACE_SINGLESTEP, pid, 0, 0) < 0) { perror("singlestep"); return 3; }
        waitpid(pid, &st, 0);
        if (WIFEXITED(st)) break;
        if (WIFSTOPPED(st) && WSTOPSIG(st) == SIGSTOP) break;           /* end marker */
        struct user_regs_struct r;
        ptrace(PTRACE_GETREGS, pid, 0, &r);
        steps++;
        if (r.rip >= lo && r.rip < hi) { bump(r.rip - lo); inside++; }
    }
    ptrace(PTRACE_KILL, pid, 0, 0);
    FILE* o = fopen(argv[7], "w");
    fprintf(o, "# steps %lu inside_oracle %lu\n", (unsigned long)steps, (unsigned long)inside);
    for (uint64_t i = 0; i < (1u << HBITS); i++) if (hk[i] || hv[i]) fprintf(o, "%lx %lu\n", (unsigned long)hk[i], (unsigned long)hv[i]);
    fclose(o);
    return 0;
}
