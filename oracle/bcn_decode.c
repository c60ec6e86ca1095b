/*
 * oracle/bcn_decode.c -- TEST INFRASTRUCTURE.  From-spec BC1 / BC3 / BC7 / BC6H(unsigned)
 * block decoders, written from the format definitions (not from the encoder):
 * the readable statement of those definitions inside the reference tree is its
 * vendored decoder, 3rdParty/DirectXTex/DirectXTex/BC6HBC7.cpp (weights :35-37,
 * partitions :40, anchors :247, BC6H decode :1077-1210, Unquantize :1313-1346,
 * FinishUnquantize :1349-1359, BC7 mode table :537, BC7 decode :1937-2140) and
 * BC.cpp for BC1/BC3.  The reference's only validity check of its bitstreams is
 * that this decoder displays them (PreviewDialog path), so "every emitted block
 * decodes, and decodes close to the source" is the strongest pin the reference
 * offers; tests/test_decode_validity.py applies it to oracle and GPU output.
 *
 * Partition/anchor data: bc7_tables.h (generated from the decoder's spec-form
 * tables); BC6H header layouts: bc6h_layout.h (generated from the decoder's
 * ms_aDesc/ms_aInfo).
 */
#include <string.h>
#include "oracle.h"
#include "bc7_tables.h"
#include "bc6h_layout.h"

static const int W2[] = { 0, 21, 43, 64 };
static const int W3[] = { 0, 9, 18, 27, 37, 46, 55, 64 };
static const int W4[] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };

static uint32_t get_bits(const uint8_t* blk, int* pos, int n)
{
    uint32_t v = 0;
    for (int i = 0; i < n; i++, (*pos)++)
        v |= (uint32_t)((blk[*pos >> 3] >> (*pos & 7)) & 1) << i;
    return v;
}

/* ---- BC1 (4-colour mode and 3-colour + transparent mode) ---- */
static void decode_color_block(const uint8_t blk[8], uint8_t out[64], int allow_3color)
{
    uint32_t c0 = blk[0] | blk[1] << 8, c1 = blk[2] | blk[3] << 8;
    int pal[4][4];
    uint32_t c[2] = { c0, c1 };
    for (int i = 0; i < 2; i++) {
        int r = (c[i] >> 11) & 31, g = (c[i] >> 5) & 63, b = c[i] & 31;
        pal[i][0] = (r << 3) | (r >> 2); pal[i][1] = (g << 2) | (g >> 4); pal[i][2] = (b << 3) | (b >> 2); pal[i][3] = 255;
    }
    if (c0 > c1 || !allow_3color) {
        for (int ch = 0; ch < 3; ch++) {
            pal[2][ch] = (2 * pal[0][ch] + pal[1][ch] + 1) / 3;
            pal[3][ch] = (pal[0][ch] + 2 * pal[1][ch] + 1) / 3;
        }
        pal[2][3] = pal[3][3] = 255;
    } else {
        for (int ch = 0; ch < 3; ch++) { pal[2][ch] = (pal[0][ch] + pal[1][ch]) / 2; pal[3][ch] = 0; }
        pal[2][3] = 255; pal[3][3] = 0;
    }
    uint32_t idx = blk[4] | blk[5] << 8 | blk[6] << 16 | (uint32_t)blk[7] << 24;
    for (int k = 0; k < 16; k++) {
        int q = (idx >> (2 * k)) & 3;
        for (int ch = 0; ch < 4; ch++) out[k * 4 + ch] = (uint8_t)pal[q][ch];
    }
}

void oracle_decode_bc1(const uint8_t blk[8], uint8_t out[64]) { decode_color_block(blk, out, 1); }

/* ---- BC3 = interpolated alpha block + BC1 colour block (always 4-colour) ---- */
/* the 8-byte interpolated-scalar block of BC3 alpha / BC4 / BC5: value of texel k -> out[k * 4 + channel] */
static void decode_scalar_block(const uint8_t blk[8], uint8_t out[64], int channel)
{
    int a[8];
    a[0] = blk[0]; a[1] = blk[1];
    if (a[0] > a[1]) {
        for (int i = 1; i < 7; i++) a[1 + i] = ((7 - i) * a[0] + i * a[1] + 3) / 7;
    } else {
        for (int i = 1; i < 5; i++) a[1 + i] = ((5 - i) * a[0] + i * a[1] + 2) / 5;
        a[6] = 0; a[7] = 255;
    }
    uint64_t bits = 0;
    for (int i = 0; i < 6; i++) bits |= (uint64_t)blk[2 + i] << (8 * i);
    for (int k = 0; k < 16; k++) out[k * 4 + channel] = (uint8_t)a[(bits >> (3 * k)) & 7];
}

void oracle_decode_bc3(const uint8_t blk[16], uint8_t out[64])
{
    decode_color_block(blk + 8, out, 0);
    decode_scalar_block(blk, out, 3);
}

/* ---- BC4_UNORM / BC5_UNORM to RGBA8: (R, 0, 0, 255) / (R, G, 0, 255), 8-bit values by the integer definition ---- */
void oracle_decode_bc4_rgba8(const uint8_t blk[8], uint8_t out[64])
{
    for (int k = 0; k < 16; k++) { out[k * 4] = out[k * 4 + 1] = out[k * 4 + 2] = 0; out[k * 4 + 3] = 255; }
    decode_scalar_block(blk, out, 0);
}

void oracle_decode_bc5_rgba8(const uint8_t blk[16], uint8_t out[64])
{
    oracle_decode_bc4_rgba8(blk, out);
    decode_scalar_block(blk + 8, out, 1);
}

/* ---- BC7 ---- */
typedef struct { int ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; } bc7_mode_info;
static const bc7_mode_info BC7_MODES[8] = {
    /* subsets, partition bits, rotation bits, index-selection bit, colour bits, alpha bits,
       per-endpoint p-bit, shared p-bit, index bits, secondary index bits */
    { 3, 4, 0, 0, 4, 0, 1, 0, 3, 0 },
    { 2, 6, 0, 0, 6, 0, 0, 1, 3, 0 },
    { 3, 6, 0, 0, 5, 0, 0, 0, 2, 0 },
    { 2, 6, 0, 0, 7, 0, 1, 0, 2, 0 },
    { 1, 0, 2, 1, 5, 6, 0, 0, 2, 3 },
    { 1, 0, 2, 0, 7, 8, 0, 0, 2, 2 },
    { 1, 0, 0, 0, 7, 7, 1, 0, 4, 0 },
    { 2, 6, 0, 0, 5, 5, 1, 0, 2, 0 },
};

static const int* weights(int bits) { return bits == 2 ? W2 : (bits == 3 ? W3 : W4); }

int oracle_decode_bc7(const uint8_t blk[16], uint8_t out[64])
{
    int pos = 0, mode = 0;
    while (mode < 8 && !get_bits(blk, &pos, 1)) mode++;
    if (mode == 8) { memset(out, 0, 64); return -1; }
    const bc7_mode_info* mi = &BC7_MODES[mode];

    int shape = (int)get_bits(blk, &pos, mi->pb);
    int rot = (int)get_bits(blk, &pos, mi->rb);
    int isel = (int)get_bits(blk, &pos, mi->isb);

    int ep[6][4];      /* [subset*2 + which][channel] */
    for (int ch = 0; ch < 3; ch++)
        for (int e = 0; e < mi->ns * 2; e++) ep[e][ch] = (int)get_bits(blk, &pos, mi->cb);
    for (int e = 0; e < mi->ns * 2; e++) ep[e][3] = mi->ab ? (int)get_bits(blk, &pos, mi->ab) : 255;

    int cbits = mi->cb, abits = mi->ab;
    if (mi->epb) {
        for (int e = 0; e < mi->ns * 2; e++) {
            int p = (int)get_bits(blk, &pos, 1);
            for (int ch = 0; ch < 3; ch++) ep[e][ch] = (ep[e][ch] << 1) | p;
            if (mi->ab) ep[e][3] = (ep[e][3] << 1) | p;
        }
        cbits++; if (mi->ab) abits++;
    } else if (mi->spb) {
        for (int s = 0; s < mi->ns; s++) {
            int p = (int)get_bits(blk, &pos, 1);
            for (int e = 2 * s; e < 2 * s + 2; e++)
                for (int ch = 0; ch < 3; ch++) ep[e][ch] = (ep[e][ch] << 1) | p;
        }
        cbits++;
    }
    for (int e = 0; e < mi->ns * 2; e++) {
        for (int ch = 0; ch < 3; ch++) { int v = ep[e][ch] << (8 - cbits); ep[e][ch] = v | (v >> cbits); }
        if (mi->ab) { int v = ep[e][3] << (8 - abits); ep[e][3] = v | (v >> abits); }
    }

    int table = (mi->ns == 3) ? 64 + shape : shape;
    uint32_t pattern = (mi->ns == 1) ? 0 : BCN_PATTERN[table];
    int anchors[3] = { 0, 0, 0 };
    if (mi->ns >= 2) { anchors[1] = BCN_ANCHORS[table] >> 4; anchors[2] = BCN_ANCHORS[table] & 15; }

    int idx1[16], idx2[16];
    for (int k = 0; k < 16; k++) {
        int sub = (pattern >> (2 * k)) & 3;
        int n = mi->ib - ((k == anchors[sub]) ? 1 : 0);
        idx1[k] = (int)get_bits(blk, &pos, n);
    }
    for (int k = 0; k < 16; k++) {
        idx2[k] = 0;
        if (mi->ib2) idx2[k] = (int)get_bits(blk, &pos, mi->ib2 - (k == 0 ? 1 : 0));
    }

    for (int k = 0; k < 16; k++) {
        int sub = (pattern >> (2 * k)) & 3;
        const int* e0 = ep[2 * sub], * e1 = ep[2 * sub + 1];
        int ci = idx1[k], cbw = mi->ib, ai = idx1[k], abw = mi->ib;
        if (mi->ib2) {
            if (isel) { ci = idx2[k]; cbw = mi->ib2; }
            else      { ai = idx2[k]; abw = mi->ib2; }
        }
        int px[4];
        int wc = weights(cbw)[ci], wa = weights(abw)[ai];
        for (int ch = 0; ch < 3; ch++) px[ch] = (e0[ch] * (64 - wc) + e1[ch] * wc + 32) >> 6;
        px[3] = (e0[3] * (64 - wa) + e1[3] * wa + 32) >> 6;
        if (!mi->ab) px[3] = 255;
        if (rot) { int t = px[3]; px[3] = px[rot - 1]; px[rot - 1] = t; }
        for (int ch = 0; ch < 4; ch++) out[k * 4 + ch] = (uint8_t)px[ch];
    }
    return (pos == 128) ? mode : -2;      /* every mode consumes exactly 128 bits */
}

/* ---- BC6H, unsigned ---- */
static int bc6h_unquantize(int comp, int bits)
{
    if (bits >= 15) return comp;
    if (comp == 0) return 0;
    if (comp == ((1 << bits) - 1)) return 0xFFFF;
    return ((comp << 16) + 0x8000) >> bits;
}

static int sign_extend(int v, int bits)
{
    return (v & (1 << (bits - 1))) ? (v | ~((1 << bits) - 1)) : v;
}

int oracle_decode_bc6h(const uint8_t blk[16], uint16_t out[48])
{
    int pos = 0;
    int m = (int)get_bits(blk, &pos, 2);
    if (m >= 2) m |= (int)get_bits(blk, &pos, 3) << 2;
    int mode = -1;
    for (int i = 0; i < 14; i++) if (BC6H_LAYOUT[i].prefix == m) mode = i;
    if (mode < 0) { memset(out, 0, 96); return -1; }
    const bc6h_mode_layout* L = &BC6H_LAYOUT[mode];

    int e[4][3];     /* W, X, Y, Z  x  R, G, B */
    memset(e, 0, sizeof e);
    int shape = 0;
    const int header = L->two_regions ? 82 : 65;
    while (pos < header) {
        int cur = pos;
        if (get_bits(blk, &pos, 1)) {
            int field = L->slot[cur] >> 4, bit = L->slot[cur] & 15;
            if (field == 2) shape |= 1 << bit;
            else if (field >= 3) { int f = field - 3; e[f & 3][f >> 2] |= 1 << bit; }
            else { memset(out, 0, 96); return -3; }   /* a mode bit slot inside the payload: malformed table */
        }
    }

    if (L->transformed) {
        for (int ch = 0; ch < 3; ch++) {
            int mask = (1 << L->base_bits[ch]) - 1;
            for (int k = 1; k < (L->two_regions ? 4 : 2); k++)
                e[k][ch] = (sign_extend(e[k][ch], L->delta_bits[ch]) + e[0][ch]) & mask;
        }
    }

    const int ib = L->two_regions ? 3 : 4;
    const int* w = L->two_regions ? W3 : W4;
    uint32_t pattern = L->two_regions ? BCN_PATTERN[shape] : 0;
    int anchor1 = L->two_regions ? (BCN_ANCHORS[shape] >> 4) : -1;

    for (int k = 0; k < 16; k++) {
        int region = (pattern >> (2 * k)) & 3;
        int n = ib - ((k == 0 || (region == 1 && k == anchor1)) ? 1 : 0);
        int idx = (int)get_bits(blk, &pos, n);
        for (int ch = 0; ch < 3; ch++) {
            int a = bc6h_unquantize(e[region * 2][ch], L->base_bits[ch]);
            int b = bc6h_unquantize(e[region * 2 + 1][ch], L->base_bits[ch]);
            int v = (a * (64 - w[idx]) + b * w[idx] + 32) >> 6;
            v = (v * 31) >> 6;
            if (v > 0x7BFF) v = 0x7BFF;
            out[ch * 16 + k] = (uint16_t)v;
        }
    }
    return (pos == 128) ? mode : -2;
}
