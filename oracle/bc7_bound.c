/* oracle/bc7_bound.c -- TEST INFRASTRUCTURE (never linked or called by the product).
 *
 * CPU restatement of the product's lower bound of a two-subset BC7 encoding (intel-texture-works-plugin_amd/csrc/bc7_exact.hpp
 * two_subset_bound), used by its bounded mode order: modes 1 and 3 are run last and only for the blocks where some shape's bound is
 * still below what the other modes achieved.  The reference has no such function; what pins it is the property the order relies on --
 * for every block and every shape the bound does not exceed the error of ANY mode 1 / 3 encoding of that shape, in particular
 * bc7_enc_mode01237_part_fast's (kernel.ispc:1279-1297) and every refinement step's (:1329-1356).  tests/test_bc7_bound.py checks that
 * against the oracle's own errors, and the device function against this file.
 *
 * Why it is a bound: a subset decodes to levels floor(L + 1/2) per channel, L on the segment between the integer endpoints
 * (kernel.ispc:1164-1170), i.e. to points within sqrt(3)/2 of one line.  With d_t = distance of texel t to that line, the subset's error is
 * sum (d_t - sqrt(3)/2)_+^2 >= (sqrt(sum d_t^2) - sqrt(3)/2 sqrt(n))_+^2 and sum d_t^2 >= R = trace - lambda_max of the subset's scatter
 * matrix (the best line of all).  lambda_max <= ||M^2||_F^(1/2) for the (scaled) scatter matrix M. */
#include <math.h>
#include <stdint.h>
#include "oracle.h"
#include "bc7_tables.h"

static float residual_bound_n(const int32_t m[6], const int32_t s[3], int32_t n)      /* n x residual, from below */
{
    const int32_t c00 = n * m[0] - s[0] * s[0], c01 = n * m[1] - s[0] * s[1], c02 = n * m[2] - s[0] * s[2];
    const int32_t c11 = n * m[3] - s[1] * s[1], c12 = n * m[4] - s[1] * s[2], c22 = n * m[5] - s[2] * s[2];
    const float t = (float)(c00 + c11 + c22);
    const float inv = 1.0f / fmaxf(t, 1.0f);
    const float a = (float)c00 * inv, b = (float)c01 * inv, c = (float)c02 * inv, d = (float)c11 * inv, e = (float)c12 * inv, f = (float)c22 * inv;
    const float bb = b * b, cc = c * c, ee = e * e;
    const float A = a * a + bb + cc, B = a * b + b * d + c * e, C = a * c + b * e + c * f;
    const float D = bb + d * d + ee, E = b * c + d * e + e * f, F = cc + ee + f * f;
    const float off = B * B + C * C + E * E;
    const float fro = (A * A + D * D + F * F) + (off + off);
    const float lam = sqrtf(sqrtf(fro));                                     /* ||M^2||_F^(1/2) >= largest eigenvalue / trace */
    const float r = ((a + d + f) - lam) - 1e-5f;
    return fmaxf(r, 0.0f) * t;
}

/* block: planar floats as everywhere in the oracle (block[16 * channel + texel], integers 0..255); shape 0..63 */
float oracle_bc7_two_subset_bound(const float block[64], int shape)
{
    const uint32_t mask0 = BCN_SUBSET_MASKS[shape] & 0xffffu;
    int32_t m[2][6] = {{0}}, s[2][3] = {{0}}, n[2] = {0, 0};
    for (int k = 0; k < 16; k++) {
        const int j = (mask0 >> k) & 1u ? 0 : 1;
        const int32_t r = (int32_t)block[k], g = (int32_t)block[16 + k], b = (int32_t)block[32 + k];
        m[j][0] += r * r; m[j][1] += r * g; m[j][2] += r * b; m[j][3] += g * g; m[j][4] += g * b; m[j][5] += b * b;
        s[j][0] += r; s[j][1] += g; s[j][2] += b;
        n[j]++;
    }
    float total = 0.f;
    for (int j = 0; j < 2; j++) {
        const float slack = (float)(0.8660254037844386 * sqrt((double)n[j]) * (1.0 + 1e-6));
        const float dj = sqrtf(residual_bound_n(m[j], s[j], n[j]) * (1.0f / (float)n[j]) * 0.9999999f) * 0.999999f - slack;
        const float ej = fmaxf(dj, 0.0f);
        total += ej * ej;
    }
    return total * 0.999999f;
}

/* Round 5: the same bound for ONE segment through the whole block (csrc/bc7.hip mode6_cannot_win).  Under an RGB profile (channels == 3) a
 * mode 6 encoding decodes every texel to floor(L + 1/2) per colour channel with L on the segment between its two endpoints
 * (kernel.ispc:1657-1689 through block_quant, :1133-1193): rounded points of one line in RGB, so its error is at least
 * (sqrt(R) - sqrt(3)/2 sqrt(16))_+^2 with R the residual of the 16 texels about their best line.  Mode 6 replaces the block only on a strict
 * `<` (:1684), so where this number has been reached by the modes before it, mode 6 need not run. */
float oracle_bc7_one_line_bound(const float block[64])
{
    int32_t m[6] = {0}, s[3] = {0};
    for (int k = 0; k < 16; k++) {
        const int32_t r = (int32_t)block[k], g = (int32_t)block[16 + k], b = (int32_t)block[32 + k];
        m[0] += r * r; m[1] += r * g; m[2] += r * b; m[3] += g * g; m[4] += g * b; m[5] += b * b;
        s[0] += r; s[1] += g; s[2] += b;
    }
    const float slack = (float)(0.8660254037844386 * 4.0 * (1.0 + 1e-6));
    const float d = sqrtf(residual_bound_n(m, s, 16) * (1.0f / 16.0f) * 0.9999999f) * 0.999999f - slack;
    const float e = fmaxf(d, 0.0f);
    return e * e * 0.999999f;
}
