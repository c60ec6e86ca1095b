/*
 * oracle/bc7.c -- TEST INFRASTRUCTURE.  Scalar restatement of the BC7 encoder,
 * kernel.ispc:616-2037, one function per reference function, same evaluation
 * order, under the pinned arithmetic of x86_math.h.  The helpers marked
 * "shared" are also what the BC6H restatement (bc6h.c) calls, exactly as
 * kernel.ispc:2174-2300, 2982-3031 reuse them.
 *
 * Division lowering follows the catalogue in SURVEY.md section 8c (S2):
 *   x / const -> x * (1.f/const);  x / y -> x * rcp(y);  `proj /= div`
 *   (kernel.ispc:1158, compound assignment) stays an IEEE divide.
 * Storage the reference leaves uninitialised is zero-initialised here (S10):
 *   best_qep / best_qblock / best_data, and the alpha slots of `ep` in the
 *   3-channel refine loop, which ep_quant0367 reads when state->channels == 4
 *   (kernel.ispc:1333-1343 with 1013-1016).
 */
#include "bc_common.h"
#include "bc7_shared.h"
#include "bc7_tables.h"

/* ---------------------------------------------------------------- tables */

/* kernel.ispc:675-686 */
const int32_t* get_unquant_table(int bits)
{
    static const int32_t t2[] = { 0, 21, 43, 64 };
    static const int32_t t3[] = { 0, 9, 18, 27, 37, 46, 55, 64 };
    static const int32_t t4[] = { 0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64 };
    return bits == 2 ? t2 : (bits == 3 ? t3 : t4);
}

/* kernel.ispc:688-710 */
uint32_t get_pattern(int part_id) { return BCN_PATTERN[part_id]; }

/* kernel.ispc:712-739 */
int32_t get_pattern_mask(int part_id, int j)
{
    uint32_t mask_packed = BCN_SUBSET_MASKS[part_id];
    int32_t mask0 = (int32_t)(mask_packed & 0xFFFF);
    int32_t mask1 = (int32_t)(mask_packed >> 16);
    return (j == 2) ? (~mask0) & (~mask1) : ((j == 0) ? mask0 : mask1);
}

/* kernel.ispc:741-758 */
void get_skips(int32_t skips[3], int part_id)
{
    int32_t skip_packed = BCN_ANCHORS[part_id];
    skips[0] = 0;
    skips[1] = skip_packed >> 4;
    skips[2] = skip_packed & 15;
}

/* ------------------------------------------------------------ PCA helpers */

/* kernel.ispc:763-803 (shared) */
void compute_stats_masked(float stats[15], const float block[64], int32_t mask, int channels)
{
    for (int i = 0; i < 15; i++) stats[i] = 0;

    int32_t mask_shifted = (int32_t)((uint32_t)mask << 1);
    for (int k = 0; k < 16; k++) {
        mask_shifted >>= 1;
        int32_t flag = (mask_shifted & 1);

        float rgba[4];
        for (int p = 0; p < channels; p++) rgba[p] = block[k + p * 16];
        for (int p = 0; p < channels; p++) rgba[p] *= (float)flag;
        stats[14] += (float)flag;

        stats[10] += rgba[0];
        stats[11] += rgba[1];
        stats[12] += rgba[2];

        stats[0] += rgba[0] * rgba[0];
        stats[1] += rgba[0] * rgba[1];
        stats[2] += rgba[0] * rgba[2];

        stats[4] += rgba[1] * rgba[1];
        stats[5] += rgba[1] * rgba[2];

        stats[7] += rgba[2] * rgba[2];

        if (channels == 4) {
            stats[13] += rgba[3];
            stats[3] += rgba[0] * rgba[3];
            stats[6] += rgba[1] * rgba[3];
            stats[8] += rgba[2] * rgba[3];
            stats[9] += rgba[3] * rgba[3];
        }
    }
}

/* kernel.ispc:805-823 (shared): every `a*b/n` is (a*b)*rcp(n) */
void covar_from_stats(float covar[10], const float stats[15], int channels)
{
    const float rn = ispc_rcp(stats[14]);
    covar[0] = stats[0] - stats[10 + 0] * stats[10 + 0] * rn;
    covar[1] = stats[1] - stats[10 + 0] * stats[10 + 1] * rn;
    covar[2] = stats[2] - stats[10 + 0] * stats[10 + 2] * rn;

    covar[4] = stats[4] - stats[10 + 1] * stats[10 + 1] * rn;
    covar[5] = stats[5] - stats[10 + 1] * stats[10 + 2] * rn;

    covar[7] = stats[7] - stats[10 + 2] * stats[10 + 2] * rn;

    if (channels == 4) {
        covar[3] = stats[3] - stats[10 + 0] * stats[10 + 3] * rn;
        covar[6] = stats[6] - stats[10 + 1] * stats[10 + 3] * rn;
        covar[8] = stats[8] - stats[10 + 2] * stats[10 + 3] * rn;
        covar[9] = stats[9] - stats[10 + 3] * stats[10 + 3] * rn;
    }
}

/* kernel.ispc:825-832 */
static void compute_covar_dc_masked(float covar[10], float dc[4], const float block[64], int32_t mask, int channels)
{
    float stats[15];
    compute_stats_masked(stats, block, mask, channels);
    covar_from_stats(covar, stats, channels);
    const float rn = ispc_rcp(stats[14]);
    for (int p = 0; p < channels; p++) dc[p] = stats[10 + p] * rn;
}

/* kernel.ispc:834-855 */
static void block_pca_axis(float axis[4], float dc[4], const float block[64], int32_t mask, int channels)
{
    const int powerIterations = 8;

    float covar[10];
    /* covar[3,6,8,9] are never written for channels == 3 but are scaled below; keep them defined */
    for (int k = 0; k < 10; k++) covar[k] = 0;
    compute_covar_dc_masked(covar, dc, block, mask, channels);

    const float inv_var = 1.0f / (256 * 256);
    for (int k = 0; k < 10; k++) covar[k] *= inv_var;

    const float eps = sqf(0.001f);
    covar[0] += eps;
    covar[4] += eps;
    covar[7] += eps;
    covar[9] += eps;

    compute_axis(axis, covar, powerIterations, channels);
}

/* kernel.ispc:857-894 (shared; BC6H uses this unclamped form) */
void block_segment_core(float ep[], const float block[64], int32_t mask, int channels)
{
    float axis[4], dc[4];
    block_pca_axis(axis, dc, block, mask, channels);

    float ext[2];
    ext[0] = +INFINITY;           /* +1e99 as a float literal */
    ext[1] = -INFINITY;

    int32_t mask_shifted = (int32_t)((uint32_t)mask << 1);
    for (int k = 0; k < 16; k++) {
        mask_shifted >>= 1;
        if ((mask_shifted & 1) == 0) continue;

        float dot = 0;
        for (int p = 0; p < channels; p++)
            dot += axis[p] * (block[16 * p + k] - dc[p]);

        ext[0] = fmin_x86(ext[0], dot);
        ext[1] = fmax_x86(ext[1], dot);
    }

    if (ext[1] - ext[0] < 1.0f) {
        ext[0] -= 0.5f;
        ext[1] += 0.5f;
    }

    for (int i = 0; i < 2; i++)
    for (int p = 0; p < channels; p++)
        ep[4 * i + p] = ext[i] * axis[p] + dc[p];
}

/* kernel.ispc:896-905 */
static void block_segment(float ep[], const float block[64], int32_t mask, int channels)
{
    block_segment_core(ep, block, mask, channels);
    for (int i = 0; i < 2; i++)
    for (int p = 0; p < channels; p++)
        ep[4 * i + p] = fclamp_x86(ep[4 * i + p], 0.0f, 255.0f);
}

/* kernel.ispc:907-939 -- modifies covar in place, like the reference */
static float get_pca_bound(float covar[10], int channels)
{
    const int powerIterations = 4;

    const float inv_var = 1.0f / (256 * 256);
    for (int k = 0; k < 10; k++) covar[k] *= inv_var;

    const float eps = sqf(0.001f);
    covar[0] += eps;
    covar[4] += eps;
    covar[7] += eps;

    float axis[4];
    compute_axis(axis, covar, powerIterations, channels);

    float vec[4];
    if (channels == 3) ssymm3(vec, covar, axis);
    if (channels == 4) ssymm4(vec, covar, axis);

    float sq_sum = 0.f;
    for (int p = 0; p < channels; p++) sq_sum += sqf(vec[p]);
    float lambda = sqrtf(sq_sum);

    float bound = covar[0] + covar[4] + covar[7];
    if (channels == 4) bound += covar[9];
    bound -= lambda;
    bound = fmax_x86(bound, 0.0f);
    return bound;
}

/* kernel.ispc:952-971 (shared) */
float block_pca_bound_split(const float block[64], int32_t mask, const float full_stats[15], int channels)
{
    float stats[15];
    compute_stats_masked(stats, block, mask, channels);

    float covar1[10];
    for (int k = 0; k < 10; k++) covar1[k] = 0;
    covar_from_stats(covar1, stats, channels);

    for (int i = 0; i < 15; i++) stats[i] = full_stats[i] - stats[i];

    float covar2[10];
    for (int k = 0; k < 10; k++) covar2[k] = 0;
    covar_from_stats(covar2, stats, channels);

    float bound = 0.f;
    bound += get_pca_bound(covar1, channels);
    bound += get_pca_bound(covar2, channels);

    return sqrtf(bound) * 256;
}

/* --------------------------------------------------- endpoint quantisation */

/* kernel.ispc:976-981 */
static int32_t unpack_to_byte(int32_t v, int bits)
{
    int32_t vv = v << (8 - bits);
    return vv + (int32_t)((uint32_t)vv >> bits);
}

#define INV255 (1.0f / 255.0f)      /* x/255f -> x*(1.f/255.f), 0x3b808081 */

/* kernel.ispc:983-1022 */
static void ep_quant0367(int32_t qep[], const float ep[], int mode, int channels)
{
    int bits = 7;
    if (mode == 0) bits = 4;
    if (mode == 7) bits = 5;

    int levels = 1 << bits;
    int levels2 = levels * 2 - 1;

    for (int i = 0; i < 2; i++) {
        int32_t qep_b[8];

        for (int b = 0; b < 2; b++)
        for (int p = 0; p < 4; p++) {
            int32_t v = f2i_x86((ep[i * 4 + p] * INV255 * (float)levels2 - (float)b) * 0.5f + 0.5f) * 2 + b;
            qep_b[b * 4 + p] = iclamp(v, b, levels2 - 1 + b);
        }

        float ep_b[8];
        for (int j = 0; j < 8; j++) ep_b[j] = (float)qep_b[j];

        if (mode == 0)
            for (int j = 0; j < 8; j++) ep_b[j] = (float)unpack_to_byte(qep_b[j], 5);

        float err0 = 0.f, err1 = 0.f;
        for (int p = 0; p < channels; p++) {
            err0 += sqf(ep[i * 4 + p] - ep_b[0 + p]);
            err1 += sqf(ep[i * 4 + p] - ep_b[4 + p]);
        }

        for (int p = 0; p < 4; p++)
            qep[i * 4 + p] = (err0 < err1) ? qep_b[0 + p] : qep_b[4 + p];
    }
}

/* kernel.ispc:1024-1052 */
static void ep_quant1(int32_t qep[], const float ep[], int mode)
{
    (void)mode;
    int32_t qep_b[16];

    for (int b = 0; b < 2; b++)
    for (int i = 0; i < 8; i++) {
        int32_t v = f2i_x86((ep[i] * INV255 * 127.f - (float)b) * 0.5f + 0.5f) * 2 + b;
        qep_b[b * 8 + i] = iclamp(v, b, 126 + b);
    }

    float ep_b[16];
    for (int k = 0; k < 16; k++) ep_b[k] = (float)unpack_to_byte(qep_b[k], 7);

    float err0 = 0.f, err1 = 0.f;
    for (int j = 0; j < 2; j++)
    for (int p = 0; p < 3; p++) {
        err0 += sqf(ep[j * 4 + p] - ep_b[0 + j * 4 + p]);
        err1 += sqf(ep[j * 4 + p] - ep_b[8 + j * 4 + p]);
    }

    for (int i = 0; i < 8; i++)
        qep[i] = (err0 < err1) ? qep_b[0 + i] : qep_b[8 + i];
}

/* kernel.ispc:1054-1065 */
static void ep_quant245(int32_t qep[], const float ep[], int mode)
{
    int bits = 5;
    if (mode == 5) bits = 7;
    int levels = 1 << bits;

    for (int i = 0; i < 8; i++) {
        int32_t v = f2i_x86(ep[i] * INV255 * (float)(levels - 1) + 0.5f);
        qep[i] = iclamp(v, 0, levels - 1);
    }
}

static const int pairs_table[] = { 3, 2, 3, 2, 1, 1, 1, 2 };

/* kernel.ispc:1067-1091 */
static void ep_quant(int32_t qep[], const float ep[], int mode, int channels)
{
    const int pairs = pairs_table[mode];
    if (mode == 0 || mode == 3 || mode == 6 || mode == 7) {
        for (int i = 0; i < pairs; i++) ep_quant0367(&qep[i * 8], &ep[i * 8], mode, channels);
    } else if (mode == 1) {
        for (int i = 0; i < pairs; i++) ep_quant1(&qep[i * 8], &ep[i * 8], mode);
    } else if (mode == 2 || mode == 4 || mode == 5) {
        for (int i = 0; i < pairs; i++) ep_quant245(&qep[i * 8], &ep[i * 8], mode);
    }
}

/* kernel.ispc:1093-1122 */
static void ep_dequant(float ep[], const int32_t qep[], int mode)
{
    const int pairs = pairs_table[mode];
    if (mode == 3 || mode == 6) {
        for (int i = 0; i < 8 * pairs; i++) ep[i] = (float)qep[i];
    } else if (mode == 1 || mode == 5) {
        for (int i = 0; i < 8 * pairs; i++) ep[i] = (float)unpack_to_byte(qep[i], 7);
    } else if (mode == 0 || mode == 2 || mode == 4) {
        for (int i = 0; i < 8 * pairs; i++) ep[i] = (float)unpack_to_byte(qep[i], 5);
    } else if (mode == 7) {
        for (int i = 0; i < 8 * pairs; i++) ep[i] = (float)unpack_to_byte(qep[i], 6);
    }
}

/* kernel.ispc:1124-1128 */
static void ep_quant_dequant(int32_t qep[], float ep[], int mode, int channels)
{
    ep_quant(qep, ep, mode, channels);
    ep_dequant(ep, qep, mode);
}

/* ------------------------------------------------------ pixel quantisation */

/* kernel.ispc:1133-1193 (shared) */
float block_quant(uint32_t qblock[2], const float block[64], int bits, const float ep[], uint32_t pattern, int channels)
{
    float total_err = 0;
    const int32_t* unquant_table = get_unquant_table(bits);
    int32_t levels = 1 << bits;

    for (int k = 0; k < 2; k++) qblock[k] = 0;

    int32_t pattern_shifted = (int32_t)pattern;
    for (int k = 0; k < 16; k++) {
        int32_t j = pattern_shifted & 3;
        pattern_shifted >>= 2;

        float proj = 0;
        float div = 0;
        for (int p = 0; p < channels; p++) {
            float ep_a = ep[8 * j + 0 + p];
            float ep_b = ep[8 * j + 4 + p];
            proj += (block[k + p * 16] - ep_a) * (ep_b - ep_a);
            div += sqf(ep_b - ep_a);
        }

        proj = ORACLE_DIV_1158(proj, div);            /* :1158 true IEEE divide in the pinned model (x86_math.h) */

        int32_t q1 = f2i_x86(proj * (float)levels + 0.5f);
        q1 = iclamp(q1, 1, levels - 1);

        float err0 = 0, err1 = 0;
        int32_t w0 = unquant_table[q1 - 1];
        int32_t w1 = unquant_table[q1];

        for (int p = 0; p < channels; p++) {
            float ep_a = ep[8 * j + 0 + p];
            float ep_b = ep[8 * j + 4 + p];
            float dec_v0 = (float)f2i_x86(((float)(64 - w0) * ep_a + (float)w0 * ep_b + 32) * 0.015625f);
            float dec_v1 = (float)f2i_x86(((float)(64 - w1) * ep_a + (float)w1 * ep_b + 32) * 0.015625f);
            err0 += sqf(dec_v0 - block[k + p * 16]);
            err1 += sqf(dec_v1 - block[k + p * 16]);
        }

        int32_t best_err = f2i_x86(err1);             /* :1178 float error truncated to int */
        int32_t best_q = q1;
        if (err0 < err1) {
            best_err = f2i_x86(err0);
            best_q = q1 - 1;
        }

        qblock[k / 8] += ((uint32_t)best_q) << (4 * (k % 8));
        total_err += (float)best_err;
    }

    return total_err;
}

/* ------------------------------------------------ LS endpoint refinement */

/* kernel.ispc:1198-1262 (shared) */
void opt_endpoints(float ep[], const float block[64], int bits, const uint32_t qblock[2], int32_t mask, int channels)
{
    int levels = 1 << bits;

    float Atb1[4] = { 0, 0, 0, 0 };
    float sum_q = 0;
    float sum_qq = 0;
    float sum[5] = { 0, 0, 0, 0, 0 };

    int32_t mask_shifted = (int32_t)((uint32_t)mask << 1);
    for (int k1 = 0; k1 < 2; k1++) {
        uint32_t qbits_shifted = qblock[k1];
        for (int k2 = 0; k2 < 8; k2++) {
            int k = k1 * 8 + k2;
            float q = (float)(int32_t)(qbits_shifted & 15);
            qbits_shifted >>= 4;

            mask_shifted >>= 1;
            if ((mask_shifted & 1) == 0) continue;

            /* `int x = (levels-1)-q;` : float expression truncated to int, then used as float */
            int32_t x = f2i_x86((float)(levels - 1) - q);

            sum_q += q;
            sum_qq += q * q;

            sum[4] += 1;
            for (int p = 0; p < channels; p++) sum[p] += block[k + p * 16];
            for (int p = 0; p < channels; p++) Atb1[p] += (float)x * block[k + p * 16];
        }
    }

    float Atb2[4];
    for (int p = 0; p < channels; p++)
        Atb2[p] = (float)(levels - 1) * sum[p] - Atb1[p];

    float Cxx = sum[4] * sqf((float)(levels - 1)) - (float)(2 * (levels - 1)) * sum_q + sum_qq;
    float Cyy = sum_qq;
    float Cxy = (float)(levels - 1) * sum_q - sum_qq;
    float scale = (float)(levels - 1) * ispc_rcp(Cxx * Cyy - Cxy * Cxy);     /* :1242 */

    for (int p = 0; p < channels; p++) {
        ep[0 + p] = (Atb1[p] * Cyy - Atb2[p] * Cxy) * scale;
        ep[4 + p] = (Atb2[p] * Cxx - Atb1[p] * Cxy) * scale;
    }

    if (fabsf(Cxx * Cyy - Cxy * Cxy) < 0.001f) {
        /* flatten */
        const float rn = ispc_rcp(sum[4]);                                   /* :1258 sum/sum[4] */
        for (int p = 0; p < channels; p++) {
            ep[0 + p] = sum[p] * rn;
            ep[4 + p] = ep[0 + p];
        }
    }
}

/* ------------------------------------------------------ mode search */

typedef struct {
    float block[64];
    float opaque_err;
    float best_err;
    uint32_t best_data[5];

    int mode_selection[4];
    int refineIterations[8];
    int skip_mode2;
    int fastSkipTreshold_mode1, fastSkipTreshold_mode3, fastSkipTreshold_mode7;
    int mode45_channel0;
    int refineIterations_channel;
    int channels;
} bc7_enc_state;

typedef struct {
    int32_t qep[8];
    uint32_t qblock[2];
    int32_t aqep[2];
    uint32_t aqblock[2];
    int32_t rotation;
    int32_t swap;
} mode45_parameters;

static void bc7_code_mode01237(uint32_t data[5], int32_t qep[], uint32_t qblock[2], int part_id, int mode);
static void bc7_code_mode45(uint32_t data[5], mode45_parameters* params, int mode);
static void bc7_code_mode6(uint32_t data[5], int32_t qep[8], uint32_t qblock[2]);

/* kernel.ispc:1267-1277 */
static float compute_opaque_err(const float block[64], int channels)
{
    if (channels == 3) return 0;
    float err = 0.f;
    for (int k = 0; k < 16; k++) err += sqf(block[48 + k] - 255);
    return err;
}

/* kernel.ispc:1279-1297 */
static float bc7_enc_mode01237_part_fast(int32_t qep[24], uint32_t qblock[2], const float block[64], int part_id, int mode)
{
    uint32_t pattern = get_pattern(part_id);
    int bits = 2;  if (mode == 0 || mode == 1) bits = 3;
    int pairs = 2; if (mode == 0 || mode == 2) pairs = 3;
    int channels = 3; if (mode == 7) channels = 4;

    float ep[24];
    for (int i = 0; i < 24; i++) ep[i] = 0;           /* S10: alpha slots are read by ep_quant for 3-channel modes */
    for (int j = 0; j < pairs; j++) {
        int32_t mask = get_pattern_mask(part_id, j);
        block_segment(&ep[j * 8], block, mask, channels);
    }

    ep_quant_dequant(qep, ep, mode, channels);

    return block_quant(qblock, block, bits, ep, pattern, channels);
}

/* kernel.ispc:1299-1363 */
static void bc7_enc_mode01237(bc7_enc_state* state, int mode, const int32_t part_list[], int part_count)
{
    if (part_count == 0) return;
    int bits = 2;  if (mode == 0 || mode == 1) bits = 3;
    int pairs = 2; if (mode == 0 || mode == 2) pairs = 3;
    int channels = 3; if (mode == 7) channels = 4;

    int32_t best_qep[24];
    uint32_t best_qblock[2] = { 0, 0 };
    int32_t best_part_id = -1;
    float best_err = INFINITY;
    for (int i = 0; i < 24; i++) best_qep[i] = 0;

    for (int part = 0; part < part_count; part++) {
        int32_t part_id = part_list[part] & 63;
        if (pairs == 3) part_id += 64;

        int32_t qep[24];
        uint32_t qblock[2];
        float err = bc7_enc_mode01237_part_fast(qep, qblock, state->block, part_id, mode);

        if (err < best_err) {
            for (int i = 0; i < 8 * pairs; i++) best_qep[i] = qep[i];
            for (int k = 0; k < 2; k++) best_qblock[k] = qblock[k];
            best_part_id = part_id;
            best_err = err;
        }
    }

    /* Every candidate error NaN is impossible for 8-bit input (errors are sums of ints), so a winner exists;
     * guard the table lookups anyway (the reference would index with -1). */
    if (best_part_id < 0) best_part_id = (pairs == 3) ? 64 : 0;

    int refineIterations = state->refineIterations[mode];
    for (int it = 0; it < refineIterations; it++) {
        float ep[24];
        for (int i = 0; i < 24; i++) ep[i] = 0;        /* S10 */
        for (int j = 0; j < pairs; j++) {
            int32_t mask = get_pattern_mask(best_part_id, j);
            opt_endpoints(&ep[j * 8], state->block, bits, best_qblock, mask, channels);
        }

        int32_t qep[24];
        uint32_t qblock[2];

        ep_quant_dequant(qep, ep, mode, state->channels);   /* :1343 passes state->channels, not the local */

        uint32_t pattern = get_pattern(best_part_id);
        float err = block_quant(qblock, state->block, bits, ep, pattern, channels);

        if (err < best_err) {
            for (int i = 0; i < 8 * pairs; i++) best_qep[i] = qep[i];
            for (int k = 0; k < 2; k++) best_qblock[k] = qblock[k];
            best_err = err;
        }
    }

    if (mode != 7) best_err += state->opaque_err;

    if (best_err < state->best_err) {
        state->best_err = best_err;
        bc7_code_mode01237(state->best_data, best_qep, best_qblock, best_part_id, mode);
    }
}

/* kernel.ispc:1365-1384 (shared) */
void partial_sort_list(int32_t list[], int length, int partial_count)
{
    for (int k = 0; k < partial_count; k++) {
        int best_idx = k;
        int32_t best_value = list[k];
        for (int i = k + 1; i < length; i++) {
            if (best_value > list[i]) {
                best_value = list[i];
                best_idx = i;
            }
        }
        list[best_idx] = list[k];
        list[k] = best_value;
    }
}

/* kernel.ispc:1386-1394 */
static void bc7_enc_mode02(bc7_enc_state* state)
{
    int32_t part_list[64];
    for (int part = 0; part < 64; part++) part_list[part] = part;

    bc7_enc_mode01237(state, 0, part_list, 16);
    if (!state->skip_mode2) bc7_enc_mode01237(state, 2, part_list, 64);
}

static int imax2(int a, int b) { return a > b ? a : b; }

/* kernel.ispc:1396-1415 */
static void bc7_enc_mode13(bc7_enc_state* state)
{
    if (state->fastSkipTreshold_mode1 == 0 && state->fastSkipTreshold_mode3 == 0) return;

    float full_stats[15];
    compute_stats_masked(full_stats, state->block, -1, 3);

    int32_t part_list[64];
    for (int part = 0; part < 64; part++) {
        int32_t mask = get_pattern_mask(part + 0, 0);
        float bound12 = block_pca_bound_split(state->block, mask, full_stats, 3);
        int32_t bound = f2i_x86(bound12);
        part_list[part] = (int32_t)((uint32_t)part + (uint32_t)bound * 64u);
    }

    partial_sort_list(part_list, 64, imax2(state->fastSkipTreshold_mode1, state->fastSkipTreshold_mode3));
    bc7_enc_mode01237(state, 1, part_list, state->fastSkipTreshold_mode1);
    bc7_enc_mode01237(state, 3, part_list, state->fastSkipTreshold_mode3);
}

/* kernel.ispc:1417-1435 */
static void bc7_enc_mode7(bc7_enc_state* state)
{
    if (state->fastSkipTreshold_mode7 == 0) return;

    float full_stats[15];
    compute_stats_masked(full_stats, state->block, -1, state->channels);

    int32_t part_list[64];
    for (int part = 0; part < 64; part++) {
        int32_t mask = get_pattern_mask(part + 0, 0);
        float bound12 = block_pca_bound_split(state->block, mask, full_stats, state->channels);
        int32_t bound = f2i_x86(bound12);
        part_list[part] = (int32_t)((uint32_t)part + (uint32_t)bound * 64u);
    }

    partial_sort_list(part_list, 64, state->fastSkipTreshold_mode7);
    bc7_enc_mode01237(state, 7, part_list, state->fastSkipTreshold_mode7);
}

/* kernel.ispc:1437-1447 */
static void channel_quant_dequant(int32_t qep[2], float ep[2], int epbits)
{
    int32_t elevels = (1 << epbits);
    for (int i = 0; i < 2; i++) {
        int32_t v = f2i_x86(ep[i] * INV255 * (float)(elevels - 1) + 0.5f);
        qep[i] = iclamp(v, 0, elevels - 1);
        ep[i] = (float)unpack_to_byte(qep[i], epbits);
    }
}

/* kernel.ispc:1449-1496 */
static void channel_opt_endpoints(float ep[2], const float block[16], int bits, const uint32_t qblock[2])
{
    int levels = 1 << bits;

    float Atb1 = 0;
    float sum_q = 0;
    float sum_qq = 0;
    float sum = 0;

    for (int k1 = 0; k1 < 2; k1++) {
        uint32_t qbits_shifted = qblock[k1];
        for (int k2 = 0; k2 < 8; k2++) {
            int k = k1 * 8 + k2;
            float q = (float)(int32_t)(qbits_shifted & 15);
            qbits_shifted >>= 4;

            int32_t x = f2i_x86((float)(levels - 1) - q);

            sum_q += q;
            sum_qq += q * q;

            sum += block[k];
            Atb1 += (float)x * block[k];
        }
    }

    float Atb2 = (float)(levels - 1) * sum - Atb1;

    float Cxx = 16 * sqf((float)(levels - 1)) - (float)(2 * (levels - 1)) * sum_q + sum_qq;
    float Cyy = sum_qq;
    float Cxy = (float)(levels - 1) * sum_q - sum_qq;
    float scale = (float)(levels - 1) * ispc_rcp(Cxx * Cyy - Cxy * Cxy);     /* :1483 */

    ep[0] = (Atb1 * Cyy - Atb2 * Cxy) * scale;
    ep[1] = (Atb2 * Cxx - Atb1 * Cxy) * scale;

    ep[0] = fclamp_x86(ep[0], 0.0f, 255.0f);
    ep[1] = fclamp_x86(ep[1], 0.0f, 255.0f);

    if (fabsf(Cxx * Cyy - Cxy * Cxy) < 0.001f) {
        ep[0] = sum * 0.0625f;                                               /* :1493 sum/16 */
        ep[1] = ep[0];
    }
}

/* kernel.ispc:1498-1538 */
static float channel_opt_quant(uint32_t qblock[2], const float block[16], int bits, const float ep[])
{
    const int32_t* unquant_table = get_unquant_table(bits);
    int32_t levels = (1 << bits);

    qblock[0] = 0;
    qblock[1] = 0;

    float total_err = 0;

    for (int k = 0; k < 16; k++) {
        float proj = (block[k] - ep[0]) * ispc_rcp(ep[1] - ep[0] + 0.001f);  /* :1510 binary divide -> rcp */

        int32_t q1 = f2i_x86(proj * (float)levels + 0.5f);
        q1 = iclamp(q1, 1, levels - 1);

        float err0 = 0, err1 = 0;
        int32_t w0 = unquant_table[q1 - 1];
        int32_t w1 = unquant_table[q1];

        float dec_v0 = (float)f2i_x86(((float)(64 - w0) * ep[0] + (float)w0 * ep[1] + 32) * 0.015625f);
        float dec_v1 = (float)f2i_x86(((float)(64 - w1) * ep[0] + (float)w1 * ep[1] + 32) * 0.015625f);
        err0 += sqf(dec_v0 - block[k]);
        err1 += sqf(dec_v1 - block[k]);

        int32_t best_err = f2i_x86(err1);
        int32_t best_q = q1;
        if (err0 < err1) {
            best_err = f2i_x86(err0);
            best_q = q1 - 1;
        }

        qblock[k / 8] += ((uint32_t)best_q) << (4 * (k % 8));
        total_err += (float)best_err;
    }

    return total_err;
}

/* kernel.ispc:1540-1563 */
static float opt_channel(bc7_enc_state* state, uint32_t qblock[2], int32_t qep[2], const float block[16], int bits, int epbits)
{
    float ep[2] = { 255, 0 };

    for (int k = 0; k < 16; k++) {
        ep[0] = fmin_x86(ep[0], block[k]);
        ep[1] = fmax_x86(ep[1], block[k]);
    }

    channel_quant_dequant(qep, ep, epbits);
    float err = channel_opt_quant(qblock, block, bits, ep);

    const int refineIterations = state->refineIterations_channel;
    for (int i = 0; i < refineIterations; i++) {
        channel_opt_endpoints(ep, block, bits, qblock);
        channel_quant_dequant(qep, ep, epbits);
        err = channel_opt_quant(qblock, block, bits, ep);
    }

    return err;
}

/* kernel.ispc:1565-1621 */
static void bc7_enc_mode45_candidate(bc7_enc_state* state, mode45_parameters* best_candidate,
                                     float* best_err, int mode, int rotation, int swap)
{
    int bits = 2;
    int abits = 2;   if (mode == 4) abits = 3;
    int aepbits = 8; if (mode == 4) aepbits = 6;
    if (swap == 1) { bits = 3; abits = 2; }

    float block[48];
    for (int k = 0; k < 16; k++) {
        for (int p = 0; p < 3; p++) block[k + p * 16] = state->block[k + p * 16];

        if (rotation < 3) {
            if (state->channels == 4) block[k + rotation * 16] = state->block[k + 3 * 16];
            if (state->channels == 3) block[k + rotation * 16] = 255;
        }
    }

    float ep[8];
    for (int i = 0; i < 8; i++) ep[i] = 0;              /* S10: slots 3/7 feed only unused qep[3], qep[7] */
    block_segment(ep, block, -1, 3);

    int32_t qep[8];
    ep_quant_dequant(qep, ep, mode, 3);

    uint32_t qblock[2];
    float err = block_quant(qblock, block, bits, ep, 0, 3);

    int refineIterations = state->refineIterations[mode];
    for (int i = 0; i < refineIterations; i++) {
        opt_endpoints(ep, block, bits, qblock, -1, 3);
        ep_quant_dequant(qep, ep, mode, 3);
        err = block_quant(qblock, block, bits, ep, 0, 3);
    }

    int32_t aqep[2];
    uint32_t aqblock[2];
    err += opt_channel(state, aqblock, aqep, &state->block[rotation * 16], abits, aepbits);

    if (err < *best_err) {
        swap_ints(best_candidate->qep, qep, 8);
        swap_uints(best_candidate->qblock, qblock, 2);
        swap_ints(best_candidate->aqep, aqep, 2);
        swap_uints(best_candidate->aqblock, aqblock, 2);
        best_candidate->rotation = rotation;
        best_candidate->swap = swap;
        *best_err = err;
    }
}

/* kernel.ispc:1623-1655 */
static void bc7_enc_mode45(bc7_enc_state* state)
{
    mode45_parameters best_candidate;
    float best_err = state->best_err;

    memset(&best_candidate, 0, sizeof(mode45_parameters));

    int channel0 = state->mode45_channel0;
    for (int p = channel0; p < state->channels; p++) {
        bc7_enc_mode45_candidate(state, &best_candidate, &best_err, 4, p, 0);
        bc7_enc_mode45_candidate(state, &best_candidate, &best_err, 4, p, 1);
    }

    if (best_err < state->best_err) {
        state->best_err = best_err;
        bc7_code_mode45(state->best_data, &best_candidate, 4);
    }

    for (int p = channel0; p < state->channels; p++)
        bc7_enc_mode45_candidate(state, &best_candidate, &best_err, 5, p, 0);

    if (best_err < state->best_err) {
        state->best_err = best_err;
        bc7_code_mode45(state->best_data, &best_candidate, 5);
    }
}

/* kernel.ispc:1657-1689 */
static void bc7_enc_mode6(bc7_enc_state* state)
{
    int mode = 6;
    int bits = 4;
    float ep[8];
    for (int i = 0; i < 8; i++) ep[i] = 0;
    block_segment(ep, state->block, -1, state->channels);

    if (state->channels == 3) ep[3] = ep[7] = 255;

    int32_t qep[8];
    ep_quant_dequant(qep, ep, mode, state->channels);

    uint32_t qblock[2];
    float err = block_quant(qblock, state->block, bits, ep, 0, state->channels);

    int refineIterations = state->refineIterations[mode];
    for (int i = 0; i < refineIterations; i++) {
        opt_endpoints(ep, state->block, bits, qblock, -1, state->channels);
        ep_quant_dequant(qep, ep, mode, state->channels);
        err = block_quant(qblock, state->block, bits, ep, 0, state->channels);
    }

    if (err < state->best_err) {
        state->best_err = err;
        bc7_code_mode6(state->best_data, qep, qblock);
    }
}

/* ------------------------------------------------------ bitstream coding */

/* kernel.ispc:1694-1706 (shared) */
void bc7_code_apply_swap_mode456(int32_t qep[], int channels, uint32_t qblock[2], int bits)
{
    int levels = 1 << bits;
    if ((qblock[0] & 15) >= (uint32_t)(levels / 2)) {
        swap_ints(&qep[0], &qep[channels], channels);
        for (int k = 0; k < 2; k++)
            qblock[k] = (uint32_t)(0x11111111u * (uint32_t)(levels - 1)) - qblock[k];
    }
}

/* kernel.ispc:1708-1733 (shared) */
int32_t bc7_code_apply_swap_mode01237(int32_t qep[], uint32_t qblock[2], int mode, int part_id)
{
    int bits = 2;  if (mode == 0 || mode == 1) bits = 3;
    int pairs = 2; if (mode == 0 || mode == 2) pairs = 3;

    int32_t flips = 0;
    int levels = 1 << bits;
    int32_t skips[3];
    get_skips(skips, part_id);

    for (int j = 0; j < pairs; j++) {
        int32_t k0 = skips[j];
        int32_t q = (int32_t)((qblock[k0 >> 3] << (28 - (k0 & 7) * 4)) >> 28);

        if (q >= levels / 2) {
            swap_ints(&qep[8 * j], &qep[8 * j + 4], 4);
            uint32_t pmask = (uint32_t)get_pattern_mask(part_id, j);
            flips |= (int32_t)pmask;
        }
    }
    return flips;
}

/* kernel.ispc:1735-1744 (shared) */
void put_bits(uint32_t data[5], int* pos, int bits, int32_t v)
{
    data[*pos / 32] |= ((uint32_t)v) << (*pos % 32);
    if (*pos % 32 + bits > 32)
        data[*pos / 32 + 1] |= ((uint32_t)v) >> (32 - *pos % 32);
    *pos += bits;
}

/* kernel.ispc:1746-1765 */
static void data_shl_1bit_from(uint32_t data[5], int32_t from)
{
    if (from < 96) {
        uint32_t shifted = (data[2] >> 1) | (data[3] << 31);
        uint32_t mask = (uint32_t)((int32_t)(((uint32_t)1 << (from - 64)) - 1u) >> 1);
        data[2] = (mask & data[2]) | (~mask & shifted);
        data[3] = (data[3] >> 1) | (data[4] << 31);
        data[4] = data[4] >> 1;
    } else if (from < 128) {
        uint32_t shifted = (data[3] >> 1) | (data[4] << 31);
        uint32_t mask = (uint32_t)((int32_t)(((uint32_t)1 << (from - 96)) - 1u) >> 1);
        data[3] = (mask & data[3]) | (~mask & shifted);
        data[4] = data[4] >> 1;
    }
}

/* kernel.ispc:1767-1785 (shared) */
void bc7_code_qblock(uint32_t data[5], int* pPos, const uint32_t qblock[2], int bits, int32_t flips)
{
    int levels = 1 << bits;
    int32_t flips_shifted = flips;
    for (int k1 = 0; k1 < 2; k1++) {
        uint32_t qbits_shifted = qblock[k1];
        for (int k2 = 0; k2 < 8; k2++) {
            int32_t q = (int32_t)(qbits_shifted & 15);
            if ((flips_shifted & 1) > 0) q = (levels - 1) - q;

            if (k1 == 0 && k2 == 0) put_bits(data, pPos, bits - 1, q);
            else                    put_bits(data, pPos, bits, q);
            qbits_shifted >>= 4;
            flips_shifted >>= 1;
        }
    }
}

/* kernel.ispc:1787-1805 (shared) */
void bc7_code_adjust_skip_mode01237(uint32_t data[5], int mode, int part_id)
{
    int bits = 2;  if (mode == 0 || mode == 1) bits = 3;
    int pairs = 2; if (mode == 0 || mode == 2) pairs = 3;

    int32_t skips[3];
    get_skips(skips, part_id);

    if (pairs > 2 && skips[1] < skips[2]) {
        int32_t t = skips[1]; skips[1] = skips[2]; skips[2] = t;
    }

    for (int j = 1; j < pairs; j++) {
        int32_t k = skips[j];
        data_shl_1bit_from(data, 128 + (pairs - 1) - (15 - k) * bits);
    }
}

/* kernel.ispc:1807-1877 */
static void bc7_code_mode01237(uint32_t data[5], int32_t qep[], uint32_t qblock[2], int part_id, int mode)
{
    int bits = 2;  if (mode == 0 || mode == 1) bits = 3;
    int pairs = 2; if (mode == 0 || mode == 2) pairs = 3;
    int channels = 3; if (mode == 7) channels = 4;

    int32_t flips = bc7_code_apply_swap_mode01237(qep, qblock, mode, part_id);

    for (int k = 0; k < 5; k++) data[k] = 0;
    int pos = 0;

    put_bits(data, &pos, mode + 1, 1 << mode);

    if (mode == 0) put_bits(data, &pos, 4, part_id & 15);
    else           put_bits(data, &pos, 6, part_id & 63);

    for (int p = 0; p < channels; p++)
    for (int j = 0; j < pairs * 2; j++) {
        if (mode == 0)      put_bits(data, &pos, 4, qep[j * 4 + 0 + p] >> 1);
        else if (mode == 1) put_bits(data, &pos, 6, qep[j * 4 + 0 + p] >> 1);
        else if (mode == 2) put_bits(data, &pos, 5, qep[j * 4 + 0 + p]);
        else if (mode == 3) put_bits(data, &pos, 7, qep[j * 4 + 0 + p] >> 1);
        else if (mode == 7) put_bits(data, &pos, 5, qep[j * 4 + 0 + p] >> 1);
    }

    if (mode == 1)
        for (int j = 0; j < 2; j++) put_bits(data, &pos, 1, qep[j * 8] & 1);

    if (mode == 0 || mode == 3 || mode == 7)
        for (int j = 0; j < pairs * 2; j++) put_bits(data, &pos, 1, qep[j * 4] & 1);

    bc7_code_qblock(data, &pos, qblock, bits, flips);
    bc7_code_adjust_skip_mode01237(data, mode, part_id);
}

/* kernel.ispc:1879-1939 */
static void bc7_code_mode45(uint32_t data[5], mode45_parameters* params, int mode)
{
    int32_t qep[8];
    uint32_t qblock[2];
    int32_t aqep[2];
    uint32_t aqblock[2];

    /* the reference swaps the winner out of *params into (uninitialised) locals; what is left behind in
     * *params is never coded, so a copy is an exact model */
    memcpy(qep, params->qep, sizeof qep);
    memcpy(qblock, params->qblock, sizeof qblock);
    memcpy(aqep, params->aqep, sizeof aqep);
    memcpy(aqblock, params->aqblock, sizeof aqblock);
    int32_t rotation = params->rotation;
    int32_t swap = params->swap;

    int bits = 2;
    int abits = 2;   if (mode == 4) abits = 3;
    int epbits = 7;  if (mode == 4) epbits = 5;
    int aepbits = 8; if (mode == 4) aepbits = 6;

    if (!swap) {
        bc7_code_apply_swap_mode456(qep, 4, qblock, bits);
        bc7_code_apply_swap_mode456(aqep, 1, aqblock, abits);
    } else {
        swap_uints(qblock, aqblock, 2);
        bc7_code_apply_swap_mode456(aqep, 1, qblock, bits);
        bc7_code_apply_swap_mode456(qep, 4, aqblock, abits);
    }

    for (int k = 0; k < 5; k++) data[k] = 0;
    int pos = 0;

    put_bits(data, &pos, mode + 1, 1 << mode);
    put_bits(data, &pos, 2, (rotation + 1) & 3);
    if (mode == 4) put_bits(data, &pos, 1, swap);

    for (int p = 0; p < 3; p++) {
        put_bits(data, &pos, epbits, qep[0 + p]);
        put_bits(data, &pos, epbits, qep[4 + p]);
    }

    put_bits(data, &pos, aepbits, aqep[0]);
    put_bits(data, &pos, aepbits, aqep[1]);

    bc7_code_qblock(data, &pos, qblock, bits, 0);
    bc7_code_qblock(data, &pos, aqblock, abits, 0);
}

/* kernel.ispc:1941-1964 */
static void bc7_code_mode6(uint32_t data[5], int32_t qep[8], uint32_t qblock[2])
{
    bc7_code_apply_swap_mode456(qep, 4, qblock, 4);

    for (int k = 0; k < 5; k++) data[k] = 0;
    int pos = 0;

    put_bits(data, &pos, 7, 64);

    for (int p = 0; p < 4; p++) {
        put_bits(data, &pos, 7, qep[0 + p] >> 1);
        put_bits(data, &pos, 7, qep[4 + p] >> 1);
    }

    put_bits(data, &pos, 1, qep[0] & 1);
    put_bits(data, &pos, 1, qep[4] & 1);

    bc7_code_qblock(data, &pos, qblock, 4, 0);
}

/* ------------------------------------------------------------- core */

/* kernel.ispc:1970-1977 */
static void CompressBlockBC7_core(bc7_enc_state* state)
{
    if (state->mode_selection[0]) bc7_enc_mode02(state);
    if (state->mode_selection[1]) bc7_enc_mode13(state);
    if (state->mode_selection[1]) bc7_enc_mode7(state);
    if (state->mode_selection[2]) bc7_enc_mode45(state);
    if (state->mode_selection[3]) bc7_enc_mode6(state);
}

/* kernel.ispc:1979-2012 */
static void bc7_enc_copy_settings(bc7_enc_state* state, const oracle_bc7_settings* settings)
{
    state->channels = settings->channels;
    state->mode_selection[0] = settings->mode_selection[0];
    state->skip_mode2 = settings->skip_mode2;
    state->refineIterations[0] = settings->refineIterations[0];
    state->refineIterations[2] = settings->refineIterations[2];
    state->mode_selection[1] = settings->mode_selection[1];
    state->fastSkipTreshold_mode1 = settings->fastSkipTreshold_mode1;
    state->fastSkipTreshold_mode3 = settings->fastSkipTreshold_mode3;
    state->fastSkipTreshold_mode7 = settings->fastSkipTreshold_mode7;
    state->refineIterations[1] = settings->refineIterations[1];
    state->refineIterations[3] = settings->refineIterations[3];
    state->refineIterations[7] = settings->refineIterations[7];
    state->mode_selection[2] = settings->mode_selection[2];
    state->mode45_channel0 = settings->mode45_channel0;
    state->refineIterations_channel = settings->refineIterations_channel;
    state->refineIterations[4] = settings->refineIterations[4];
    state->refineIterations[5] = settings->refineIterations[5];
    state->mode_selection[3] = settings->mode_selection[3];
    state->refineIterations[6] = settings->refineIterations[6];
}

void oracle_bc7_block(const float block[64], const oracle_bc7_settings* settings, uint32_t data[4], float* best_err)
{
    bc7_enc_state state;
    memset(&state, 0, sizeof state);
    bc7_enc_copy_settings(&state, settings);
    memcpy(state.block, block, sizeof state.block);
    state.best_err = INFINITY;
    state.opaque_err = compute_opaque_err(state.block, state.channels);
    CompressBlockBC7_core(&state);
    for (int k = 0; k < 4; k++) data[k] = state.best_data[k];
    if (best_err) *best_err = state.best_err;
}

/* Study hook (tools/variants/bc7_bound_study.py, tests): bc7_enc_mode01237_part_fast's error for every shape of one mode, and the
 * sort key bc7_enc_mode13 / mode7 order the shapes by (kernel.ispc:1403-1410; 0..63 for modes 0 / 2). */
void oracle_bc7_part_fast_errors(const float block[64], int mode, float err[64], int32_t key[64])
{
    const int pairs = (mode == 0 || mode == 2) ? 3 : 2;
    const int channels = mode == 7 ? 4 : 3;
    const int count = mode == 0 ? 16 : 64;
    float full_stats[15];
    compute_stats_masked(full_stats, block, -1, channels);
    for (int part = 0; part < 64; part++) {
        err[part] = -1.0f;
        key[part] = part;
        if (pairs == 2) {
            float bound12 = block_pca_bound_split(block, get_pattern_mask(part, 0), full_stats, channels);
            key[part] = (int32_t)((uint32_t)part + (uint32_t)f2i_x86(bound12) * 64u);
        }
        if (part >= count) continue;
        int32_t qep[24];
        uint32_t qblock[2];
        err[part] = bc7_enc_mode01237_part_fast(qep, qblock, block, part + (pairs == 3 ? 64 : 0), mode);
    }
}

/* kernel.ispc:2014-2037 */
void oracle_CompressBlocksBC7(const oracle_surface* src, uint8_t* dst, const oracle_bc7_settings* settings)
{
    for (int yy = 0; yy < src->height / 4; yy++)
    for (int xx = 0; xx < src->width / 4; xx++) {
        float block[64];
        uint32_t data[4];
        load_block_interleaved_rgba(block, src, xx, yy);
        oracle_bc7_block(block, settings, data, 0);
        store_data(dst, src->width, xx, yy, data, 4);
    }
}
