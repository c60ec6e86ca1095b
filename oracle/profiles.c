/*
 * oracle/profiles.c -- TEST INFRASTRUCTURE.  Quality presets restated from
 * ispc_texcomp.cpp:20-410 as data rows.  A row writes exactly the fields the
 * reference function writes: the five RGB presets never touch
 * refineIterations[7] (ispc_texcomp.cpp:20-189).
 */
#include <string.h>
#include "oracle.h"

typedef struct {
    const char* name;
    int channels;
    int sel0, sel1, sel2, sel3;
    int skip2;
    int r0, r2;                 /* modes 0,2 */
    int n1, n3, n7;
    int r1, r3, r7;             /* r7 < 0: not written */
    int ch0, rch, r4, r5;
    int r6;
} bc7_row;

static const bc7_row BC7_ROWS[] = {
    /* name              ch  sel          skip2 r0 r2  n1  n3  n7  r1 r3 r7  ch0 rch r4 r5  r6 */
    { "ultrafast",        3, 0, 0, 0, 1,  1,    2, 2,  3,  1,  0,  2, 1, -1, 0,  0,  2, 2,  1 },
    { "veryfast",         3, 0, 1, 0, 1,  1,    2, 2,  3,  1,  0,  2, 1, -1, 0,  0,  2, 2,  1 },
    { "fast",             3, 0, 1, 0, 1,  1,    2, 2,  12, 4,  0,  2, 1, -1, 0,  0,  2, 2,  2 },
    { "basic",            3, 1, 1, 1, 1,  1,    2, 2,  12, 8,  0,  2, 2, -1, 0,  2,  2, 2,  2 },
    { "slow",             3, 1, 1, 1, 1,  0,    4, 4,  64, 64, 0,  4, 4, -1, 0,  4,  4, 4,  4 },
    { "alpha_ultrafast",  4, 0, 0, 1, 1,  1,    2, 2,  0,  0,  4,  1, 1,  2, 3,  1,  1, 1,  2 },
    { "alpha_veryfast",   4, 0, 1, 1, 1,  1,    2, 2,  0,  0,  4,  1, 1,  2, 3,  2,  2, 2,  2 },
    { "alpha_fast",       4, 0, 1, 1, 1,  1,    2, 2,  4,  4,  8,  1, 1,  2, 3,  2,  2, 2,  2 },
    { "alpha_basic",      4, 1, 1, 1, 1,  1,    2, 2,  12, 8,  8,  2, 2,  2, 0,  2,  2, 2,  2 },
    { "alpha_slow",       4, 1, 1, 1, 1,  0,    4, 4,  64, 64, 64, 4, 4,  4, 0,  4,  4, 4,  4 },
};

int oracle_GetProfile_bc7(const char* name, oracle_bc7_settings* s)
{
    for (unsigned i = 0; i < sizeof BC7_ROWS / sizeof BC7_ROWS[0]; i++) {
        const bc7_row* r = &BC7_ROWS[i];
        if (strcmp(r->name, name)) continue;
        s->channels = r->channels;
        s->mode_selection[0] = (uint8_t)r->sel0; s->mode_selection[1] = (uint8_t)r->sel1;
        s->mode_selection[2] = (uint8_t)r->sel2; s->mode_selection[3] = (uint8_t)r->sel3;
        s->skip_mode2 = (uint8_t)r->skip2;
        s->refineIterations[0] = r->r0; s->refineIterations[2] = r->r2;
        s->fastSkipTreshold_mode1 = r->n1; s->fastSkipTreshold_mode3 = r->n3; s->fastSkipTreshold_mode7 = r->n7;
        s->refineIterations[1] = r->r1; s->refineIterations[3] = r->r3;
        if (r->r7 >= 0) s->refineIterations[7] = r->r7;
        s->mode45_channel0 = r->ch0; s->refineIterations_channel = r->rch;
        s->refineIterations[4] = r->r4; s->refineIterations[5] = r->r5;
        s->refineIterations[6] = r->r6;
        return 0;
    }
    return -1;
}

int oracle_GetProfile_bc6h(const char* name, oracle_bc6h_settings* s)
{
    static const struct { const char* name; int slow, fast, n, r1p, r2p; } rows[] = {
        { "veryfast", 0, 1, 0, 0, 0 }, { "fast", 0, 1, 2, 0, 1 }, { "basic", 0, 0, 4, 2, 2 },
        { "slow", 1, 0, 10, 2, 2 }, { "veryslow", 1, 0, 32, 2, 2 },
    };
    for (unsigned i = 0; i < sizeof rows / sizeof rows[0]; i++) {
        if (strcmp(rows[i].name, name)) continue;
        s->slow_mode = (uint8_t)rows[i].slow; s->fast_mode = (uint8_t)rows[i].fast;
        s->fastSkipTreshold = rows[i].n;
        s->refineIterations_1p = rows[i].r1p; s->refineIterations_2p = rows[i].r2p;
        return 0;
    }
    return -1;
}
