/*
 * oracle/bc1_bc3.c -- TEST INFRASTRUCTURE.  Scalar restatement of the BC1/BC3
 * encoder, kernel.ispc:231-614.  One function per reference function, same
 * evaluation order; divisions lowered as catalogued in x86_math.h.
 */
#include "bc_common.h"

/* kernel.ispc:234-238 */
static int32_t stb__Mul8Bit(int32_t a, int32_t b)
{
    int32_t t = a * b + 128;
    return (t + (t >> 8)) >> 8;
}

/* kernel.ispc:240-243 -- result type is unsigned int16 in the reference */
static uint16_t stb__As16Bit(int32_t r, int32_t g, int32_t b)
{
    return (uint16_t)((stb__Mul8Bit(r, 31) << 11) + (stb__Mul8Bit(g, 63) << 5) + stb__Mul8Bit(b, 31));
}

/* kernel.ispc:245-248 */
static uint16_t enc_rgb565(const float c[3])
{
    return stb__As16Bit(f2i_x86(c[0]), f2i_x86(c[1]), f2i_x86(c[2]));
}

/* kernel.ispc:250-259 */
static void dec_rgb565(float c[3], int32_t p)
{
    int32_t c2 = (p >> 0) & 31;
    int32_t c1 = (p >> 5) & 63;
    int32_t c0 = (p >> 11) & 31;
    c[0] = (float)((c0 << 3) + (c0 >> 2));
    c[1] = (float)((c1 << 2) + (c1 >> 4));
    c[2] = (float)((c2 << 3) + (c2 >> 2));
}

/* kernel.ispc:274-306 */
static void pick_endpoints(float c0[3], float c1[3], const float block[48], const float axis[3], const float dc[3])
{
    float min_dot = 256 * 256;
    float max_dot = 0;

    for (int y = 0; y < 4; y++)
    for (int x = 0; x < 4; x++) {
        float dot = 0;
        for (int p = 0; p < 3; p++)
            dot += (block[p * 16 + y * 4 + x] - dc[p]) * axis[p];
        min_dot = fmin_x86(min_dot, dot);
        max_dot = fmax_x86(max_dot, dot);
    }

    if (max_dot - min_dot < 1.0f) {
        min_dot -= 0.5f;
        max_dot += 0.5f;
    }

    float norm_sq = 0;
    for (int p = 0; p < 3; p++) norm_sq += axis[p] * axis[p];

    float rnorm_sq = ispc_rcp(norm_sq);                                  /* :300 explicit rcp */
    for (int p = 0; p < 3; p++) {
        c0[p] = fclamp_x86(dc[p] + min_dot * rnorm_sq * axis[p], 0.0f, 255.0f);
        c1[p] = fclamp_x86(dc[p] + max_dot * rnorm_sq * axis[p], 0.0f, 255.0f);
    }
}

/* kernel.ispc:308-344 */
static uint32_t fast_quant(const float block[48], int32_t p0, int32_t p1)
{
    float c0[3], c1[3];
    dec_rgb565(c0, p0);
    dec_rgb565(c1, p1);

    float dir[3];
    for (int p = 0; p < 3; p++) dir[p] = c1[p] - c0[p];

    float sq_norm = 0;
    for (int p = 0; p < 3; p++) sq_norm += sqf(dir[p]);

    float rsq_norm = ispc_rcp(sq_norm);                                  /* :321 explicit rcp */

    for (int p = 0; p < 3; p++) dir[p] *= rsq_norm * 3;

    float bias = 0.5f;
    for (int p = 0; p < 3; p++) bias -= c0[p] * dir[p];

    uint32_t bits = 0;
    uint32_t scaler = 1;
    for (int k = 0; k < 16; k++) {
        float dot = 0;
        for (int p = 0; p < 3; p++) dot += block[k + p * 16] * dir[p];
        int32_t q = iclamp(f2i_x86(dot + bias), 0, 3);
        bits += (uint32_t)q * scaler;
        scaler *= 4;
    }
    return bits;
}

/* kernel.ispc:377-417 (the variant the encoder actually calls, :501) */
static void compute_covar_dc_ugly(float covar[6], float dc[3], const float block[48])
{
    for (int p = 0; p < 3; p++) {
        float acc = 0;
        for (int k = 0; k < 16; k++) acc += block[k + p * 16];
        dc[p] = acc * 0.0625f;                                           /* :384 acc/16 */
    }

    float covar0 = 0.f, covar1 = 0.f, covar2 = 0.f, covar3 = 0.f, covar4 = 0.f, covar5 = 0.f;
    for (int k = 0; k < 16; k++) {
        float rgb0 = block[k + 0 * 16] - dc[0];
        float rgb1 = block[k + 1 * 16] - dc[1];
        float rgb2 = block[k + 2 * 16] - dc[2];
        covar0 += rgb0 * rgb0;
        covar1 += rgb0 * rgb1;
        covar2 += rgb0 * rgb2;
        covar3 += rgb1 * rgb1;
        covar4 += rgb1 * rgb2;
        covar5 += rgb2 * rgb2;
    }
    covar[0] = covar0; covar[1] = covar1; covar[2] = covar2;
    covar[3] = covar3; covar[4] = covar4; covar[5] = covar5;
}

/* kernel.ispc:419-480 */
static void bc1_refine(int32_t pe[2], const float block[48], uint32_t bits, const float dc[3])
{
    float c0[3], c1[3];

    if ((bits ^ (bits * 4)) < 4) {
        /* all sixteen 2-bit indices equal */
        for (int p = 0; p < 3; p++) { c0[p] = dc[p]; c1[p] = dc[p]; }
    } else {
        float Atb1[3] = { 0, 0, 0 };
        float sum_q = 0, sum_qq = 0;
        uint32_t shifted_bits = bits;

        for (int k = 0; k < 16; k++) {
            float q = (float)(int32_t)(shifted_bits & 3);
            shifted_bits >>= 2;
            float x = 3 - q;
            sum_q += q;
            sum_qq += q * q;
            for (int p = 0; p < 3; p++) Atb1[p] += x * block[k + p * 16];
        }

        float sum[3], Atb2[3];
        for (int p = 0; p < 3; p++) {
            sum[p] = dc[p] * 16;
            Atb2[p] = 3 * sum[p] - Atb1[p];
        }

        float Cxx = 16 * sqf(3) - 2 * 3 * sum_q + sum_qq;
        float Cyy = sum_qq;
        float Cxy = 3 * sum_q - sum_qq;
        float scale = 3.f * ispc_rcp(Cxx * Cyy - Cxy * Cxy);             /* :466 explicit rcp */

        for (int p = 0; p < 3; p++) {
            c0[p] = (Atb1[p] * Cyy - Atb2[p] * Cxy) * scale;
            c1[p] = (Atb2[p] * Cxx - Atb1[p] * Cxy) * scale;
            c0[p] = fclamp_x86(c0[p], 0.0f, 255.0f);
            c1[p] = fclamp_x86(c1[p], 0.0f, 255.0f);
        }
    }

    pe[0] = enc_rgb565(c0);
    pe[1] = enc_rgb565(c1);
}

/* kernel.ispc:482-492 */
static uint32_t fix_qbits(uint32_t qbits)
{
    const uint32_t mask_01b = 0x55555555u, mask_10b = 0xAAAAAAAAu;
    uint32_t qbits0 = qbits & mask_01b;
    uint32_t qbits1 = qbits & mask_10b;
    return (qbits1 >> 1) + (qbits1 ^ (qbits0 << 1));
}

/* kernel.ispc:494-533 */
void oracle_bc1_core(const float block[48], uint32_t data[2])
{
    const int powerIterations = 4;
    const int refineIterations = 1;

    float covar[6], dc[3];
    compute_covar_dc_ugly(covar, dc, block);

    float eps = 0.001f;
    covar[0] += eps;
    covar[3] += eps;
    covar[5] += eps;

    float axis[3];
    compute_axis3(axis, covar, powerIterations);

    float c0[3], c1[3];
    pick_endpoints(c0, c1, block, axis, dc);

    int32_t p[2];
    p[0] = enc_rgb565(c0);
    p[1] = enc_rgb565(c1);
    if (p[0] < p[1]) swap_ints(&p[0], &p[1], 1);

    data[0] = ((uint32_t)1 << 16) * (uint32_t)p[1] + (uint32_t)p[0];
    data[1] = fast_quant(block, p[0], p[1]);

    for (int i = 0; i < refineIterations; i++) {
        bc1_refine(p, block, data[1], dc);
        if (p[0] < p[1]) swap_ints(&p[0], &p[1], 1);
        data[0] = ((uint32_t)1 << 16) * (uint32_t)p[1] + (uint32_t)p[0];
        data[1] = fast_quant(block, p[0], p[1]);
    }

    data[1] = fix_qbits(data[1]);
}

/* kernel.ispc:535-571 */
static void CompressBlockBC3_alpha(const float block[16], uint32_t data[2])
{
    float ep[2] = { 255, 0 };
    for (int k = 0; k < 16; k++) {
        ep[0] = fmin_x86(ep[0], block[k]);
        ep[1] = fmax_x86(ep[1], block[k]);
    }

    if (ep[0] == ep[1]) ep[1] = ep[0] + 0.1f;

    uint32_t qblock[2] = { 0, 0 };
    float scale = 7.f * ispc_rcp(ep[1] - ep[0]);                          /* :548 7f/(max-min) */

    for (int k = 0; k < 16; k++) {
        float v = block[k];
        float proj = (v - ep[0]) * scale + 0.5f;
        int32_t q = iclamp(f2i_x86(proj), 0, 7);
        q = 7 - q;
        if (q > 0) q++;
        if (q == 8) q = 1;
        qblock[k / 8] |= (uint32_t)q << ((k % 8) * 3);
    }

    data[0] = (uint32_t)(iclamp(f2i_x86(ep[0]), 0, 255) * 256 + iclamp(f2i_x86(ep[1]), 0, 255));
    data[0] |= qblock[0] << 16;
    data[1] = qblock[0] >> 16;
    data[1] |= qblock[1] << 8;
}

void oracle_bc1_block(const float block[48], uint32_t data[2]) { oracle_bc1_core(block, data); }
void oracle_bc3_alpha_block(const float alpha[16], uint32_t data[2]) { CompressBlockBC3_alpha(alpha, data); }

/* kernel.ispc:573-583, 598-605 */
void oracle_CompressBlocksBC1(const oracle_surface* src, uint8_t* dst)
{
    for (int yy = 0; yy < src->height / 4; yy++)
    for (int xx = 0; xx < src->width / 4; xx++) {
        float block[48];
        uint32_t data[2];
        load_block_interleaved(block, src, xx, yy);
        oracle_bc1_core(block, data);
        store_data(dst, src->width, xx, yy, data, 2);
    }
}

/* kernel.ispc:585-596, 607-614 */
void oracle_CompressBlocksBC3(const oracle_surface* src, uint8_t* dst)
{
    for (int yy = 0; yy < src->height / 4; yy++)
    for (int xx = 0; xx < src->width / 4; xx++) {
        float block[64];
        uint32_t data[4];
        load_block_interleaved_rgba(block, src, xx, yy);
        CompressBlockBC3_alpha(&block[48], &data[0]);
        oracle_bc1_core(block, &data[2]);
        store_data(dst, src->width, xx, yy, data, 4);
    }
}
