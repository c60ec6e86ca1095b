/*
 * oracle/bc6h.c -- TEST INFRASTRUCTURE.  Scalar restatement of the BC6H
 * (unsigned half) encoder, kernel.ispc:2039-3139, on top of the BC7 helpers of
 * bc7.c exactly as the reference layers it.  The reference has no signed
 * encoder: DXGI_FORMAT_BC6H_SF16 routes to this same code (IntelPlugin.cpp:841,
 * win32Threads.cpp:206); half values with the sign bit set are consumed as
 * integers >= 0x8000.
 *
 * Arithmetic notes: half/31*64 is (h*(1/31f))*64 (:3048);
 * ep/(256*256f-1)*(levels-1) is (ep*(1/65535f))*(levels-1) (:2145); the span
 * table is evaluated in float and truncated to int (:2094-2108); block_quant's
 * int error accumulator follows cvttps2dq and can go negative for wide blocks.
 */
#include "bc_common.h"
#include "bc7_shared.h"

typedef struct {
    float block[64];

    float best_err;
    uint32_t best_data[5];

    float rgb_bounds[6];
    float max_span;
    int32_t max_span_idx;

    int32_t mode;
    int32_t epb;
    int32_t qbounds[8];

    int slow_mode, fast_mode;
    int refineIterations_1p, refineIterations_2p, fastSkipTreshold;
} bc6h_enc_state;

static void bc6h_code_2p(uint32_t data[5], int32_t pqep[], uint32_t qblock[2], int part_id, int mode);
static void bc6h_code_1p(uint32_t data[5], int32_t qep[8], uint32_t qblock[2], int mode);

/* kernel.ispc:2080-2088 */
static int get_mode_prefix(int mode)
{
    static const int mode_prefix_table[] = { 0, 1, 2, 6, 10, 14, 18, 22, 26, 30, 3, 7, 11, 15 };
    return mode_prefix_table[mode];
}

/* kernel.ispc:2090-2111: float constant expressions, then `uniform int span = span_table[mode]` truncates.
 * Written with volatile-free float arithmetic so gcc folds them in float like ISPC does
 * (x/64 etc. are exact power-of-two scalings, so the fast-math rewrite changes nothing). */
static float get_span(int mode)
{
    static const float span_table[] = {
        0.9f * 65535.f / 64,
        0.9f * 65535.f / 4,
        0.8f * 65535.f / 256,
        -1, -1,
        0.9f * 65535.f / 32,
        0.9f * 65535.f / 16,
        -1, -1,
        65535.f,
        65535.f,
        0.95f * 65535.f / 8,
        0.95f * 65535.f / 32,
        6,
    };
    int32_t span = (int32_t)span_table[mode];
    return (float)span;
}

/* kernel.ispc:2113-2125 */
static int get_mode_bits(int mode)
{
    static const int mode_bits_table[] = { 10, 7, 11, -1, -1, 9, 8, -1, -1, 6, 10, 11, 12, 16 };
    return mode_bits_table[mode];
}

/* kernel.ispc:2130-2137 */
static int32_t unpack_to_uf16(uint32_t v, int32_t bits)
{
    if (bits >= 15) return (int32_t)v;
    if (v == 0) return 0;
    if (v == ((uint32_t)1 << bits) - 1) return 0xFFFF;
    return (int32_t)((v * 2 + 1) << (15 - bits));
}

#define INV65535 (1.0f / 65535.0f)    /* 0x37800080 */

/* kernel.ispc:2139-2148 */
static void ep_quant_bc6h(int32_t qep[], const float ep[], int32_t bits, int pairs)
{
    int32_t levels = 1 << bits;
    for (int i = 0; i < 8 * pairs; i++) {
        int32_t v = f2i_x86(ep[i] * INV65535 * (float)(levels - 1) + 0.5f);
        qep[i] = iclamp(v, 0, levels - 1);
    }
}

/* kernel.ispc:2150-2154 */
static void ep_dequant_bc6h(float ep[], const int32_t qep[], int32_t bits, int pairs)
{
    for (int i = 0; i < 8 * pairs; i++)
        ep[i] = (float)unpack_to_uf16((uint32_t)qep[i], bits);
}

/* kernel.ispc:2156-2169 */
static void ep_quant_dequant_bc6h(bc6h_enc_state* state, int32_t qep[], float ep[], int pairs)
{
    int32_t bits = state->epb;
    ep_quant_bc6h(qep, ep, bits, pairs);

    for (int i = 0; i < 2 * pairs; i++)
    for (int p = 0; p < 3; p++)
        qep[i * 4 + p] = iclamp(qep[i * 4 + p], state->qbounds[p], state->qbounds[4 + p]);

    ep_dequant_bc6h(ep, qep, bits, pairs);
}

/* kernel.ispc:2174-2193 */
static float bc6h_enc_2p_part_fast(bc6h_enc_state* state, int32_t qep[16], uint32_t qblock[2], int part_id)
{
    uint32_t pattern = get_pattern(part_id);
    const int bits = 3, pairs = 2, channels = 3;

    float ep[16];
    for (int i = 0; i < 16; i++) ep[i] = 0;          /* S10: slots 3/7/11/15 only feed unused qep slots */
    for (int j = 0; j < pairs; j++) {
        int32_t mask = get_pattern_mask(part_id, j);
        block_segment_core(&ep[j * 8], state->block, mask, channels);
    }

    ep_quant_dequant_bc6h(state, qep, ep, 2);

    return block_quant(qblock, state->block, bits, ep, pattern, channels);
}

/* kernel.ispc:2195-2255 */
static void bc6h_enc_2p_list(bc6h_enc_state* state, const int32_t part_list[], int part_count)
{
    if (part_count == 0) return;
    const int bits = 3, pairs = 2, channels = 3;

    int32_t best_qep[24];
    uint32_t best_qblock[2] = { 0, 0 };
    int32_t best_part_id = -1;
    float best_err = INFINITY;
    for (int i = 0; i < 24; i++) best_qep[i] = 0;

    for (int part = 0; part < part_count; part++) {
        int32_t part_id = part_list[part] & 31;

        int32_t qep[24];
        uint32_t qblock[2];
        float err = bc6h_enc_2p_part_fast(state, qep, qblock, part_id);

        if (err < best_err) {
            for (int i = 0; i < 8 * pairs; i++) best_qep[i] = qep[i];
            for (int k = 0; k < 2; k++) best_qblock[k] = qblock[k];
            best_part_id = part_id;
            best_err = err;
        }
    }

    /* all-NaN candidate errors cannot occur (errors are sums of converted ints); guard the table index anyway */
    if (best_part_id < 0) best_part_id = 0;

    int refineIterations = state->refineIterations_2p;
    for (int it = 0; it < refineIterations; it++) {
        float ep[24];
        for (int i = 0; i < 24; i++) ep[i] = 0;
        for (int j = 0; j < pairs; j++) {
            int32_t mask = get_pattern_mask(best_part_id, j);
            opt_endpoints(&ep[j * 8], state->block, bits, best_qblock, mask, channels);
        }

        int32_t qep[24];
        uint32_t qblock[2];
        ep_quant_dequant_bc6h(state, qep, ep, 2);

        uint32_t pattern = get_pattern(best_part_id);
        float err = block_quant(qblock, state->block, bits, ep, pattern, channels);

        if (err < best_err) {
            for (int i = 0; i < 8 * pairs; i++) best_qep[i] = qep[i];
            for (int k = 0; k < 2; k++) best_qblock[k] = qblock[k];
            best_err = err;
        }
    }

    if (best_err < state->best_err) {
        state->best_err = best_err;
        bc6h_code_2p(state->best_data, best_qep, best_qblock, best_part_id, state->mode);
    }
}

/* kernel.ispc:2257-2273 */
static void bc6h_enc_2p(bc6h_enc_state* state)
{
    float full_stats[15];
    compute_stats_masked(full_stats, state->block, -1, 3);

    int32_t part_list[32];
    for (int part = 0; part < 32; part++) {
        int32_t mask = get_pattern_mask(part, 0);
        float bound12 = block_pca_bound_split(state->block, mask, full_stats, 3);
        int32_t bound = f2i_x86(bound12);
        part_list[part] = (int32_t)((uint32_t)part + (uint32_t)bound * 64u);
    }

    partial_sort_list(part_list, 32, state->fastSkipTreshold);
    bc6h_enc_2p_list(state, part_list, state->fastSkipTreshold);
}

/* kernel.ispc:2275-2300 */
static void bc6h_enc_1p(bc6h_enc_state* state)
{
    float ep[8];
    for (int i = 0; i < 8; i++) ep[i] = 0;
    block_segment_core(ep, state->block, -1, 3);

    int32_t qep[8];
    ep_quant_dequant_bc6h(state, qep, ep, 1);

    uint32_t qblock[2];
    float err = block_quant(qblock, state->block, 4, ep, 0, 3);

    int refineIterations = state->refineIterations_1p;
    for (int i = 0; i < refineIterations; i++) {
        opt_endpoints(ep, state->block, 4, qblock, -1, 3);
        ep_quant_dequant_bc6h(state, qep, ep, 1);
        err = block_quant(qblock, state->block, 4, ep, 0, 3);
    }

    if (err < state->best_err) {
        state->best_err = err;
        bc6h_code_1p(state->best_data, qep, qblock, state->mode);
    }
}

/* kernel.ispc:2302-2314 */
static void compute_qbounds_rgb(bc6h_enc_state* state, const float rgb_span[3])
{
    float bounds[8];
    for (int i = 0; i < 8; i++) bounds[i] = 0;        /* slots 3/7 unused (S10) */
    for (int p = 0; p < 3; p++) {
        float middle = (state->rgb_bounds[p] + state->rgb_bounds[3 + p]) * 0.5f;
        bounds[p]     = middle - rgb_span[p] * 0.5f;
        bounds[4 + p] = middle + rgb_span[p] * 0.5f;
    }
    ep_quant_bc6h(state->qbounds, bounds, state->epb, 1);
}

/* kernel.ispc:2316-2320 */
static void compute_qbounds(bc6h_enc_state* state, float span)
{
    float rgb_span[3] = { span, span, span };
    compute_qbounds_rgb(state, rgb_span);
}

/* kernel.ispc:2322-2330 */
static void compute_qbounds2(bc6h_enc_state* state, float span, int32_t max_span_idx)
{
    float rgb_span[3] = { span, span, span };
    for (int p = 0; p < 3; p++)
        rgb_span[p] *= (p == max_span_idx) ? 2 : 1;
    compute_qbounds_rgb(state, rgb_span);
}

/* kernel.ispc:2332-2365 */
static void bc6h_test_mode(bc6h_enc_state* state, int mode, int enc, float margin)
{
    int mode_bits = get_mode_bits(mode);
    float span = get_span(mode);
    float max_span = state->max_span;
    int32_t max_span_idx = state->max_span_idx;

    if (max_span * margin > span) return;

    if (mode >= 10) {
        state->epb = mode_bits;
        state->mode = mode;
        compute_qbounds(state, span);
        if (enc) bc6h_enc_1p(state);
    } else if (mode <= 1 || mode == 5 || mode == 9) {
        state->epb = mode_bits;
        state->mode = mode;
        compute_qbounds(state, span);
        if (enc) bc6h_enc_2p(state);
    } else {
        state->epb = mode_bits;
        state->mode = mode + max_span_idx;
        compute_qbounds2(state, span, max_span_idx);
        if (enc) bc6h_enc_2p(state);
    }
}

/* ---------------------------------------------------- bitstream coding */

/* kernel.ispc:2370-2373 */
static int32_t bit_at(int32_t v, int pos) { return (v >> pos) & 1; }

/* kernel.ispc:2375-2390 */
static uint32_t reverse_bits(uint32_t v, int bits)
{
    if (bits == 2) return (v >> 1) + (v & 1) * 2;
    /* bits == 6 */
    v = (v & 0x5555) * 2 + ((v >> 1) & 0x5555);
    return (v >> 4) + ((v >> 2) & 3) * 4 + (v & 3) * 16;
}

/* kernel.ispc:2392-2980.  Endpoint order: qep[0..2] = region 0 A (r0 g0 b0), qep[4..6] = region 0 B (r1..),
 * qep[8..10] = region 1 A (r2..), qep[12..14] = region 1 B (r3..).  Arithmetic on uint32 so that negative
 * deltas wrap exactly like ISPC ints (S7). */
static void bc6h_pack(uint32_t packed[], const int32_t qep[], int mode)
{
    if (mode == 0) {
        int32_t pred_qep[16];
        for (int p = 0; p < 3; p++) {
            pred_qep[p] = qep[p];
            pred_qep[4 + p] = (qep[4 + p] - qep[p]) & 31;
            pred_qep[8 + p] = (qep[8 + p] - qep[p]) & 31;
            pred_qep[12 + p] = (qep[12 + p] - qep[p]) & 31;
        }

        uint32_t pqep[10];
        pqep[4] = (uint32_t)pred_qep[4] + (uint32_t)(pred_qep[8 + 1] & 15) * 64;
        pqep[5] = (uint32_t)pred_qep[5] + (uint32_t)(pred_qep[12 + 1] & 15) * 64;
        pqep[6] = (uint32_t)pred_qep[6] + (uint32_t)(pred_qep[8 + 2] & 15) * 64;

        pqep[4] += (uint32_t)bit_at(pred_qep[12 + 1], 4) << 5;
        pqep[5] += (uint32_t)bit_at(pred_qep[12 + 2], 0) << 5;
        pqep[6] += (uint32_t)bit_at(pred_qep[12 + 2], 1) << 5;

        pqep[8] = (uint32_t)pred_qep[8] + (uint32_t)bit_at(pred_qep[12 + 2], 2) * 32;
        pqep[9] = (uint32_t)pred_qep[12] + (uint32_t)bit_at(pred_qep[12 + 2], 3) * 32;

        packed[0] = (uint32_t)get_mode_prefix(0);
        packed[0] += (uint32_t)bit_at(pred_qep[8 + 1], 4) << 2;
        packed[0] += (uint32_t)bit_at(pred_qep[8 + 2], 4) << 3;
        packed[0] += (uint32_t)bit_at(pred_qep[12 + 2], 4) << 4;

        packed[1] = ((uint32_t)pred_qep[2] << 20) + ((uint32_t)pred_qep[1] << 10) + (uint32_t)pred_qep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
        packed[3] = (pqep[9] << 6) + pqep[8];
    } else if (mode == 1) {
        int32_t pred_qep[16];
        for (int p = 0; p < 3; p++) {
            pred_qep[p] = qep[p];
            pred_qep[4 + p] = (qep[4 + p] - qep[p]) & 63;
            pred_qep[8 + p] = (qep[8 + p] - qep[p]) & 63;
            pred_qep[12 + p] = (qep[12 + p] - qep[p]) & 63;
        }

        uint32_t pqep[8];
        pqep[0] = (uint32_t)pred_qep[0];
        pqep[0] += (uint32_t)bit_at(pred_qep[12 + 2], 0) << 7;
        pqep[0] += (uint32_t)bit_at(pred_qep[12 + 2], 1) << 8;
        pqep[0] += (uint32_t)bit_at(pred_qep[8 + 2], 4) << 9;

        pqep[1] = (uint32_t)pred_qep[1];
        pqep[1] += (uint32_t)bit_at(pred_qep[8 + 2], 5) << 7;
        pqep[1] += (uint32_t)bit_at(pred_qep[12 + 2], 2) << 8;
        pqep[1] += (uint32_t)bit_at(pred_qep[8 + 1], 4) << 9;

        pqep[2] = (uint32_t)pred_qep[2];
        pqep[2] += (uint32_t)bit_at(pred_qep[12 + 2], 3) << 7;
        pqep[2] += (uint32_t)bit_at(pred_qep[12 + 2], 5) << 8;
        pqep[2] += (uint32_t)bit_at(pred_qep[12 + 2], 4) << 9;

        pqep[4] = (uint32_t)pred_qep[4] + (uint32_t)(pred_qep[8 + 1] & 15) * 64;
        pqep[5] = (uint32_t)pred_qep[5] + (uint32_t)(pred_qep[12 + 1] & 15) * 64;
        pqep[6] = (uint32_t)pred_qep[6] + (uint32_t)(pred_qep[8 + 2] & 15) * 64;

        packed[0] = (uint32_t)get_mode_prefix(1);
        packed[0] += (uint32_t)bit_at(pred_qep[8 + 1], 5) << 2;
        packed[0] += (uint32_t)bit_at(pred_qep[12 + 1], 4) << 3;
        packed[0] += (uint32_t)bit_at(pred_qep[12 + 1], 5) << 4;

        packed[1] = (pqep[2] << 20) + (pqep[1] << 10) + pqep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
        packed[3] = ((uint32_t)pred_qep[12] << 6) + (uint32_t)pred_qep[8];
    } else if (mode == 2 || mode == 3 || mode == 4) {
        int32_t dqep[16];
        for (int p = 0; p < 3; p++) {
            int32_t mask = 15;
            if (p == mode - 2) mask = 31;
            dqep[p] = qep[p];
            dqep[4 + p] = (qep[4 + p] - qep[p]) & mask;
            dqep[8 + p] = (qep[8 + p] - qep[p]) & mask;
            dqep[12 + p] = (qep[12 + p] - qep[p]) & mask;
        }

        uint32_t pqep[10];
        pqep[0] = (uint32_t)dqep[0] & 1023;
        pqep[1] = (uint32_t)dqep[1] & 1023;
        pqep[2] = (uint32_t)dqep[2] & 1023;

        pqep[4] = (uint32_t)dqep[4] + (uint32_t)(dqep[8 + 1] & 15) * 64;
        pqep[5] = (uint32_t)dqep[5] + (uint32_t)(dqep[12 + 1] & 15) * 64;
        pqep[6] = (uint32_t)dqep[6] + (uint32_t)(dqep[8 + 2] & 15) * 64;

        pqep[8] = (uint32_t)dqep[8];
        pqep[9] = (uint32_t)dqep[12];

        if (mode == 2) {
            packed[0] = (uint32_t)get_mode_prefix(2);

            pqep[5] += (uint32_t)bit_at(dqep[0 + 1], 10) << 4;
            pqep[6] += (uint32_t)bit_at(dqep[0 + 2], 10) << 4;

            pqep[4] += (uint32_t)bit_at(dqep[0 + 0], 10) << 5;
            pqep[5] += (uint32_t)bit_at(dqep[12 + 2], 0) << 5;
            pqep[6] += (uint32_t)bit_at(dqep[12 + 2], 1) << 5;
            pqep[8] += (uint32_t)bit_at(dqep[12 + 2], 2) << 5;
            pqep[9] += (uint32_t)bit_at(dqep[12 + 2], 3) << 5;
        }
        if (mode == 3) {
            packed[0] = (uint32_t)get_mode_prefix(3);

            pqep[4] += (uint32_t)bit_at(dqep[0 + 0], 10) << 4;
            pqep[6] += (uint32_t)bit_at(dqep[0 + 2], 10) << 4;
            pqep[8] += (uint32_t)bit_at(dqep[12 + 2], 0) << 4;
            pqep[9] += (uint32_t)bit_at(dqep[8 + 1], 4) << 4;

            pqep[4] += (uint32_t)bit_at(dqep[12 + 1], 4) << 5;
            pqep[5] += (uint32_t)bit_at(dqep[0 + 1], 10) << 5;
            pqep[6] += (uint32_t)bit_at(dqep[12 + 2], 1) << 5;
            pqep[8] += (uint32_t)bit_at(dqep[12 + 2], 2) << 5;
            pqep[9] += (uint32_t)bit_at(dqep[12 + 2], 3) << 5;
        }
        if (mode == 4) {
            packed[0] = (uint32_t)get_mode_prefix(4);

            pqep[4] += (uint32_t)bit_at(dqep[0 + 0], 10) << 4;
            pqep[5] += (uint32_t)bit_at(dqep[0 + 1], 10) << 4;
            pqep[8] += (uint32_t)bit_at(dqep[12 + 2], 1) << 4;
            pqep[9] += (uint32_t)bit_at(dqep[12 + 2], 4) << 4;

            pqep[4] += (uint32_t)bit_at(dqep[8 + 2], 4) << 5;
            pqep[5] += (uint32_t)bit_at(dqep[12 + 2], 0) << 5;
            pqep[6] += (uint32_t)bit_at(dqep[0 + 2], 10) << 5;
            pqep[8] += (uint32_t)bit_at(dqep[12 + 2], 2) << 5;
            pqep[9] += (uint32_t)bit_at(dqep[12 + 2], 3) << 5;
        }

        packed[1] = (pqep[2] << 20) + (pqep[1] << 10) + pqep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
        packed[3] = (pqep[9] << 6) + pqep[8];
    } else if (mode == 5) {
        int32_t dqep[16];
        for (int p = 0; p < 3; p++) {
            dqep[p] = qep[p];
            dqep[4 + p] = (qep[4 + p] - qep[p]) & 31;
            dqep[8 + p] = (qep[8 + p] - qep[p]) & 31;
            dqep[12 + p] = (qep[12 + p] - qep[p]) & 31;
        }

        uint32_t pqep[10];
        pqep[0] = (uint32_t)dqep[0];
        pqep[1] = (uint32_t)dqep[1];
        pqep[2] = (uint32_t)dqep[2];
        pqep[4] = (uint32_t)dqep[4] + (uint32_t)(dqep[8 + 1] & 15) * 64;
        pqep[5] = (uint32_t)dqep[5] + (uint32_t)(dqep[12 + 1] & 15) * 64;
        pqep[6] = (uint32_t)dqep[6] + (uint32_t)(dqep[8 + 2] & 15) * 64;
        pqep[8] = (uint32_t)dqep[8];
        pqep[9] = (uint32_t)dqep[12];

        pqep[0] += (uint32_t)bit_at(dqep[8 + 2], 4) << 9;
        pqep[1] += (uint32_t)bit_at(dqep[8 + 1], 4) << 9;
        pqep[2] += (uint32_t)bit_at(dqep[12 + 2], 4) << 9;

        pqep[4] += (uint32_t)bit_at(dqep[12 + 1], 4) << 5;
        pqep[5] += (uint32_t)bit_at(dqep[12 + 2], 0) << 5;
        pqep[6] += (uint32_t)bit_at(dqep[12 + 2], 1) << 5;

        pqep[8] += (uint32_t)bit_at(dqep[12 + 2], 2) << 5;
        pqep[9] += (uint32_t)bit_at(dqep[12 + 2], 3) << 5;

        packed[0] = (uint32_t)get_mode_prefix(5);
        packed[1] = (pqep[2] << 20) + (pqep[1] << 10) + pqep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
        packed[3] = (pqep[9] << 6) + pqep[8];
    } else if (mode == 6 || mode == 7 || mode == 8) {
        int32_t dqep[16];
        for (int p = 0; p < 3; p++) {
            int32_t mask = 31;
            if (p == mode - 6) mask = 63;
            dqep[p] = qep[p];
            dqep[4 + p] = (qep[4 + p] - qep[p]) & mask;
            dqep[8 + p] = (qep[8 + p] - qep[p]) & mask;
            dqep[12 + p] = (qep[12 + p] - qep[p]) & mask;
        }

        uint32_t pqep[10];
        pqep[0] = (uint32_t)dqep[0];
        pqep[0] += (uint32_t)bit_at(dqep[8 + 2], 4) << 9;

        pqep[1] = (uint32_t)dqep[1];
        pqep[1] += (uint32_t)bit_at(dqep[8 + 1], 4) << 9;

        pqep[2] = (uint32_t)dqep[2];
        pqep[2] += (uint32_t)bit_at(dqep[12 + 2], 4) << 9;

        pqep[4] = (uint32_t)dqep[4] + (uint32_t)(dqep[8 + 1] & 15) * 64;
        pqep[5] = (uint32_t)dqep[5] + (uint32_t)(dqep[12 + 1] & 15) * 64;
        pqep[6] = (uint32_t)dqep[6] + (uint32_t)(dqep[8 + 2] & 15) * 64;

        pqep[8] = (uint32_t)dqep[8];
        pqep[9] = (uint32_t)dqep[12];

        if (mode == 6) {
            packed[0] = (uint32_t)get_mode_prefix(6);

            pqep[0] += (uint32_t)bit_at(dqep[12 + 1], 4) << 8;
            pqep[1] += (uint32_t)bit_at(dqep[12 + 2], 2) << 8;
            pqep[2] += (uint32_t)bit_at(dqep[12 + 2], 3) << 8;
            pqep[5] += (uint32_t)bit_at(dqep[12 + 2], 0) << 5;
            pqep[6] += (uint32_t)bit_at(dqep[12 + 2], 1) << 5;
        }
        if (mode == 7) {
            packed[0] = (uint32_t)get_mode_prefix(7);

            pqep[0] += (uint32_t)bit_at(dqep[12 + 2], 0) << 8;
            pqep[1] += (uint32_t)bit_at(dqep[8 + 1], 5) << 8;
            pqep[2] += (uint32_t)bit_at(dqep[12 + 1], 5) << 8;
            pqep[4] += (uint32_t)bit_at(dqep[12 + 1], 4) << 5;
            pqep[6] += (uint32_t)bit_at(dqep[12 + 2], 1) << 5;
            pqep[8] += (uint32_t)bit_at(dqep[12 + 2], 2) << 5;
            pqep[9] += (uint32_t)bit_at(dqep[12 + 2], 3) << 5;
        }
        if (mode == 8) {
            packed[0] = (uint32_t)get_mode_prefix(8);

            pqep[0] += (uint32_t)bit_at(dqep[12 + 2], 1) << 8;
            pqep[1] += (uint32_t)bit_at(dqep[8 + 2], 5) << 8;
            pqep[2] += (uint32_t)bit_at(dqep[12 + 2], 5) << 8;
            pqep[4] += (uint32_t)bit_at(dqep[12 + 1], 4) << 5;
            pqep[5] += (uint32_t)bit_at(dqep[12 + 2], 0) << 5;
            pqep[8] += (uint32_t)bit_at(dqep[12 + 2], 2) << 5;
            pqep[9] += (uint32_t)bit_at(dqep[12 + 2], 3) << 5;
        }

        packed[1] = (pqep[2] << 20) + (pqep[1] << 10) + pqep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
        packed[3] = (pqep[9] << 6) + pqep[8];
    } else if (mode == 9) {
        uint32_t pqep[10];

        pqep[0] = (uint32_t)qep[0];
        pqep[0] += (uint32_t)bit_at(qep[12 + 1], 4) << 6;
        pqep[0] += (uint32_t)bit_at(qep[12 + 2], 0) << 7;
        pqep[0] += (uint32_t)bit_at(qep[12 + 2], 1) << 8;
        pqep[0] += (uint32_t)bit_at(qep[8 + 2], 4) << 9;

        pqep[1] = (uint32_t)qep[1];
        pqep[1] += (uint32_t)bit_at(qep[8 + 1], 5) << 6;
        pqep[1] += (uint32_t)bit_at(qep[8 + 2], 5) << 7;
        pqep[1] += (uint32_t)bit_at(qep[12 + 2], 2) << 8;
        pqep[1] += (uint32_t)bit_at(qep[8 + 1], 4) << 9;

        pqep[2] = (uint32_t)qep[2];
        pqep[2] += (uint32_t)bit_at(qep[12 + 1], 5) << 6;
        pqep[2] += (uint32_t)bit_at(qep[12 + 2], 3) << 7;
        pqep[2] += (uint32_t)bit_at(qep[12 + 2], 5) << 8;
        pqep[2] += (uint32_t)bit_at(qep[12 + 2], 4) << 9;

        pqep[4] = (uint32_t)qep[4] + (uint32_t)(qep[8 + 1] & 15) * 64;
        pqep[5] = (uint32_t)qep[5] + (uint32_t)(qep[12 + 1] & 15) * 64;
        pqep[6] = (uint32_t)qep[6] + (uint32_t)(qep[8 + 2] & 15) * 64;

        packed[0] = (uint32_t)get_mode_prefix(9);
        packed[1] = (pqep[2] << 20) + (pqep[1] << 10) + pqep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
        packed[3] = ((uint32_t)qep[12] << 6) + (uint32_t)qep[8];
    } else if (mode == 10) {
        packed[0] = (uint32_t)get_mode_prefix(10);
        packed[1] = ((uint32_t)qep[2] << 20) + ((uint32_t)qep[1] << 10) + (uint32_t)qep[0];
        packed[2] = ((uint32_t)qep[6] << 20) + ((uint32_t)qep[5] << 10) + (uint32_t)qep[4];
    } else if (mode == 11) {
        int32_t dqep[8];
        for (int p = 0; p < 3; p++) {
            dqep[p] = qep[p];
            dqep[4 + p] = (qep[4 + p] - qep[p]) & 511;
        }

        uint32_t pqep[8];
        pqep[0] = (uint32_t)dqep[0] & 1023;
        pqep[1] = (uint32_t)dqep[1] & 1023;
        pqep[2] = (uint32_t)dqep[2] & 1023;

        pqep[4] = (uint32_t)dqep[4] + (uint32_t)(dqep[0] >> 10) * 512;
        pqep[5] = (uint32_t)dqep[5] + (uint32_t)(dqep[1] >> 10) * 512;
        pqep[6] = (uint32_t)dqep[6] + (uint32_t)(dqep[2] >> 10) * 512;

        packed[0] = (uint32_t)get_mode_prefix(11);
        packed[1] = (pqep[2] << 20) + (pqep[1] << 10) + pqep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
    } else if (mode == 12) {
        int32_t dqep[8];
        for (int p = 0; p < 3; p++) {
            dqep[p] = qep[p];
            dqep[4 + p] = (qep[4 + p] - qep[p]) & 255;
        }

        uint32_t pqep[8];
        pqep[0] = (uint32_t)dqep[0] & 1023;
        pqep[1] = (uint32_t)dqep[1] & 1023;
        pqep[2] = (uint32_t)dqep[2] & 1023;

        pqep[4] = (uint32_t)dqep[4] + reverse_bits((uint32_t)(dqep[0] >> 10), 2) * 256;
        pqep[5] = (uint32_t)dqep[5] + reverse_bits((uint32_t)(dqep[1] >> 10), 2) * 256;
        pqep[6] = (uint32_t)dqep[6] + reverse_bits((uint32_t)(dqep[2] >> 10), 2) * 256;

        packed[0] = (uint32_t)get_mode_prefix(12);
        packed[1] = (pqep[2] << 20) + (pqep[1] << 10) + pqep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
    } else if (mode == 13) {
        int32_t dqep[8];
        for (int p = 0; p < 3; p++) {
            dqep[p] = qep[p];
            dqep[4 + p] = (qep[4 + p] - qep[p]) & 15;
        }

        uint32_t pqep[8];
        pqep[0] = (uint32_t)dqep[0] & 1023;
        pqep[1] = (uint32_t)dqep[1] & 1023;
        pqep[2] = (uint32_t)dqep[2] & 1023;

        pqep[4] = (uint32_t)dqep[4] + reverse_bits((uint32_t)(dqep[0] >> 10), 6) * 16;
        pqep[5] = (uint32_t)dqep[5] + reverse_bits((uint32_t)(dqep[1] >> 10), 6) * 16;
        pqep[6] = (uint32_t)dqep[6] + reverse_bits((uint32_t)(dqep[2] >> 10), 6) * 16;

        packed[0] = (uint32_t)get_mode_prefix(13);
        packed[1] = (pqep[2] << 20) + (pqep[1] << 10) + pqep[0];
        packed[2] = (pqep[6] << 20) + (pqep[5] << 10) + pqep[4];
    }
}

/* kernel.ispc:2982-3010 */
static void bc6h_code_2p(uint32_t data[5], int32_t qep[], uint32_t qblock[2], int part_id, int mode)
{
    const int bits = 3;

    int32_t flips = bc7_code_apply_swap_mode01237(qep, qblock, 1, part_id);

    for (int k = 0; k < 5; k++) data[k] = 0;
    int pos = 0;

    uint32_t packed[4] = { 0, 0, 0, 0 };
    bc6h_pack(packed, qep, mode);

    put_bits(data, &pos, 5, (int32_t)packed[0]);
    put_bits(data, &pos, 30, (int32_t)packed[1]);
    put_bits(data, &pos, 30, (int32_t)packed[2]);
    put_bits(data, &pos, 12, (int32_t)packed[3]);
    put_bits(data, &pos, 5, part_id);

    bc7_code_qblock(data, &pos, qblock, bits, flips);
    bc7_code_adjust_skip_mode01237(data, 1, part_id);
}

/* kernel.ispc:3012-3031 */
static void bc6h_code_1p(uint32_t data[5], int32_t qep[8], uint32_t qblock[2], int mode)
{
    bc7_code_apply_swap_mode456(qep, 4, qblock, 4);

    for (int k = 0; k < 5; k++) data[k] = 0;
    int pos = 0;

    uint32_t packed[4] = { 0, 0, 0, 0 };
    bc6h_pack(packed, qep, mode);

    put_bits(data, &pos, 5, (int32_t)packed[0]);
    put_bits(data, &pos, 30, (int32_t)packed[1]);
    put_bits(data, &pos, 30, (int32_t)packed[2]);

    bc7_code_qblock(data, &pos, qblock, 4, 0);
}

/* ------------------------------------------------------------- core */

/* kernel.ispc:3036-3067 */
static void bc6h_setup(bc6h_enc_state* state)
{
    for (int p = 0; p < 3; p++) {
        state->rgb_bounds[p] = 0xFFFF;
        state->rgb_bounds[3 + p] = 0;
    }

    for (int p = 0; p < 3; p++)
    for (int k = 0; k < 16; k++) {
        state->block[p * 16 + k] = (state->block[p * 16 + k] * (1.0f / 31.0f)) * 64;     /* :3048 */
        state->rgb_bounds[p] = fmin_x86(state->rgb_bounds[p], state->block[p * 16 + k]);
        state->rgb_bounds[3 + p] = fmax_x86(state->rgb_bounds[3 + p], state->block[p * 16 + k]);
    }

    state->max_span = 0;
    state->max_span_idx = 0;

    float rgb_span[3] = { 0, 0, 0 };
    for (int p = 0; p < 3; p++) {
        rgb_span[p] = state->rgb_bounds[3 + p] - state->rgb_bounds[p];
        if (rgb_span[p] > state->max_span) {
            state->max_span_idx = p;
            state->max_span = rgb_span[p];
        }
    }
}

/* kernel.ispc:3069-3107 */
static void CompressBlockBC6H_core(bc6h_enc_state* state)
{
    bc6h_setup(state);

    if (state->slow_mode) {
        bc6h_test_mode(state, 0, 1, 0);
        bc6h_test_mode(state, 1, 1, 0);
        bc6h_test_mode(state, 2, 1, 0);
        bc6h_test_mode(state, 5, 1, 0);
        bc6h_test_mode(state, 6, 1, 0);
        bc6h_test_mode(state, 9, 1, 0);
        bc6h_test_mode(state, 10, 1, 0);
        bc6h_test_mode(state, 11, 1, 0);
        bc6h_test_mode(state, 12, 1, 0);
        bc6h_test_mode(state, 13, 1, 0);
    } else {
        const float inv1_2 = 1.0f / 1.2f;                                            /* :3092 `1 / 1.2` */
        if (state->fastSkipTreshold > 0) {
            bc6h_test_mode(state, 9, 0, 0);
            if (state->fast_mode) bc6h_test_mode(state, 1, 0, 1);
            bc6h_test_mode(state, 6, 0, inv1_2);
            bc6h_test_mode(state, 5, 0, inv1_2);
            bc6h_test_mode(state, 0, 0, inv1_2);
            bc6h_test_mode(state, 2, 0, 1);

            bc6h_enc_2p(state);
            if (!state->fast_mode) bc6h_test_mode(state, 1, 1, 0);
        }

        bc6h_test_mode(state, 10, 0, 0);
        bc6h_test_mode(state, 11, 0, 1);
        bc6h_test_mode(state, 12, 0, 1);
        bc6h_test_mode(state, 13, 0, 1);
        bc6h_enc_1p(state);
    }
}

void oracle_bc6h_block(const float block[64], const oracle_bc6h_settings* settings, uint32_t data[4], float* best_err)
{
    bc6h_enc_state state;
    memset(&state, 0, sizeof state);
    /* kernel.ispc:3109-3116 */
    state.slow_mode = settings->slow_mode;
    state.fast_mode = settings->fast_mode;
    state.fastSkipTreshold = settings->fastSkipTreshold;
    state.refineIterations_1p = settings->refineIterations_1p;
    state.refineIterations_2p = settings->refineIterations_2p;
    memcpy(state.block, block, sizeof state.block);
    state.best_err = INFINITY;
    CompressBlockBC6H_core(&state);
    for (int k = 0; k < 4; k++) data[k] = state.best_data[k];
    if (best_err) *best_err = state.best_err;
}

/* kernel.ispc:3118-3139 */
void oracle_CompressBlocksBC6H(const oracle_surface* src, uint8_t* dst, const oracle_bc6h_settings* settings)
{
    for (int yy = 0; yy < src->height / 4; yy++)
    for (int xx = 0; xx < src->width / 4; xx++) {
        float block[64];
        uint32_t data[4];
        load_block_interleaved_16bit(block, src, xx, yy);
        oracle_bc6h_block(block, settings, data, 0);
        store_data(dst, src->width, xx, yy, data, 4);
    }
}

/* TEST / STUDY HOOK (tools/round5/bc6h_bound_study.py; never called by the product): one two-region mode's scan as the reference runs it under
 * the slow profiles (kernel.ispc:2332-2365 -> :2257-2273 -> :2195-2216): the 32 shapes in ranked order, each with its ranking key's bound part
 * and bc6h_enc_2p_part_fast's error.  `mode` is kernel.ispc's number (0, 1, 2, 5, 6, 9); list[i] = shape at position i, bound[i] = (int)bound12,
 * err[i] = its error; returns max_span (the S5 quirk's domain: kernel.ispc:1178 overflows where a span exceeds 26 754). */
float oracle_bc6h_2p_scan(const float block[64], int mode, int32_t list[32], int32_t bound[32], float err[32])
{
    bc6h_enc_state state;
    memset(&state, 0, sizeof state);
    state.slow_mode = 1;
    state.fastSkipTreshold = 32;
    memcpy(state.block, block, sizeof state.block);
    state.best_err = INFINITY;
    bc6h_setup(&state);
    bc6h_test_mode(&state, mode, 0, 0);                /* epb, mode, qbounds; no encode */
    float full_stats[15];
    compute_stats_masked(full_stats, state.block, -1, 3);
    int32_t part_list[32];
    for (int part = 0; part < 32; part++) {
        int32_t mask = get_pattern_mask(part, 0);
        float bound12 = block_pca_bound_split(state.block, mask, full_stats, 3);
        part_list[part] = (int32_t)((uint32_t)part + (uint32_t)f2i_x86(bound12) * 64u);
    }
    partial_sort_list(part_list, 32, 32);
    for (int i = 0; i < 32; i++) {
        int32_t qep[24];
        uint32_t qblock[2];
        list[i] = part_list[i] & 31;
        bound[i] = part_list[i] >> 6;
        err[i] = bc6h_enc_2p_part_fast(&state, qep, qblock, list[i]);
    }
    return state.max_span;
}

/* the block as bc6h_setup leaves it (uf16-scaled texels), for the study's own bound */
void oracle_bc6h_setup_block(const float block[64], float out[64])
{
    bc6h_enc_state state;
    memset(&state, 0, sizeof state);
    memcpy(state.block, block, sizeof state.block);
    bc6h_setup(&state);
    memcpy(out, state.block, sizeof state.block);
}
