/*
 * oracle/bc4_bc5.c -- TEST INFRASTRUCTURE.  CPU restatement of the one encoder on the plugin's save path that is NOT
 * kernel.ispc: BC4_UNORM / BC5_UNORM go through DirectXTex (IntelPlugin.cpp:120-141 builds an RGBA8 scratch image,
 * IntelPlugin.cpp:271-273 calls DirectX::Compress(..., TEX_COMPRESS_DEFAULT, 0.5f, ...)).  Checker for
 * csrc/bc4_bc5.hip; never linked by the product.
 *
 * Followed line by line:
 *   block walk + partial-block replication   3rdParty/DirectXTex/DirectXTex/DirectXTexCompress.cpp:105-183
 *   D3DXEncodeBC4U / D3DXEncodeBC5U          3rdParty/DirectXTex/DirectXTex/BC4BC5.cpp:403-421, 481-512
 *   FindEndPointsBC4U                        BC4BC5.cpp:186-238
 *   OptimizeAlpha<false>                     3rdParty/DirectXTex/DirectXTex/BC.h:727-856
 *   FindClosestUNORM / DecodeFromIndex       BC4BC5.cpp:314-337, 50-72
 *
 * PARITY UNPINNED, for two reasons stated here and in DESIGN.md:
 *  1. the texel load is DirectXMath's XMLoadUByteN4 (a Windows SDK header, not vendored under /root/reference).  Its
 *     published SSE path scales the converted integer by the constant 1.0f/255.0f (one multiply), which is what is
 *     restated below; the scalar path of the same function divides by 255.0f.  The two differ in the last bit for
 *     some codes; both map 0 -> 0.0f and 255 -> 1.0f exactly, which is all the 6-/8-step decision needs.
 *  2. DirectXTex is built with /fp:fast (DirectXTex_Desktop_201x.vcxproj), which lets MSVC re-associate and replace
 *     divisions by reciprocal multiplies; what it actually did to this code cannot be known without the binary.
 *     This restatement evaluates every expression as written, in IEEE fp32, one rounding per operation
 *     (-ffp-contract=off), and the HIP kernel does exactly the same.
 * The reference ships no BC4/BC5 golden outputs either.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#include "oracle.h"

/* BC.h:729-732 -- compile-time fp32 quotients */
static const float kC6[6] = { 5.0f/5.0f, 4.0f/5.0f, 3.0f/5.0f, 2.0f/5.0f, 1.0f/5.0f, 0.0f/5.0f };
static const float kD6[6] = { 0.0f/5.0f, 1.0f/5.0f, 2.0f/5.0f, 3.0f/5.0f, 4.0f/5.0f, 5.0f/5.0f };
static const float kC8[8] = { 7.0f/7.0f, 6.0f/7.0f, 5.0f/7.0f, 4.0f/7.0f, 3.0f/7.0f, 2.0f/7.0f, 1.0f/7.0f, 0.0f/7.0f };
static const float kD8[8] = { 0.0f/7.0f, 1.0f/7.0f, 2.0f/7.0f, 3.0f/7.0f, 4.0f/7.0f, 5.0f/7.0f, 6.0f/7.0f, 7.0f/7.0f };

/* OptimizeAlpha<false> (BC.h:727-856): Newton iterations on the two ramp ends; steps = 8 (plain ramp) or 6 (ramp plus
 * the exact codes 0 and 1). */
static void optimize_alpha_unorm(float* px, float* py, const float* pts, int steps)
{
    const float* pc = (steps == 6) ? kC6 : kC8;
    const float* pd = (steps == 6) ? kD6 : kD8;
    const float max_value = 1.0f, min_value = 0.0f;
    float fx = max_value, fy = min_value;
    if (steps == 8) {
        for (int i = 0; i < 16; i++) {
            if (pts[i] < fx) fx = pts[i];
            if (pts[i] > fy) fy = pts[i];
        }
    } else {
        for (int i = 0; i < 16; i++) {
            if (pts[i] < fx && pts[i] > min_value) fx = pts[i];
            if (pts[i] > fy && pts[i] < max_value) fy = pts[i];
        }
        if (fx == fy) fy = max_value;
    }
    const float fsteps = (float)(steps - 1);
    for (int it = 0; it < 8; it++) {
        if ((fy - fx) < (1.0f / 256.0f)) break;
        const float scale = fsteps / (fy - fx);
        float ps[8];
        for (int s = 0; s < steps; s++) ps[s] = pc[s] * fx + pd[s] * fy;
        if (steps == 6) { ps[6] = min_value; ps[7] = max_value; }
        float dx = 0.0f, dy = 0.0f, d2x = 0.0f, d2y = 0.0f;
        for (int i = 0; i < 16; i++) {
            const float dot = (pts[i] - fx) * scale;
            int s;
            if (dot <= 0.0f) s = ((steps == 6) && (pts[i] <= fx * 0.5f)) ? 6 : 0;
            else if (dot >= fsteps) s = ((steps == 6) && (pts[i] >= (fy + 1.0f) * 0.5f)) ? 7 : (steps - 1);
            else s = (int32_t)(dot + 0.5f);
            if (s < steps) {
                const float diff = ps[s] - pts[i];
                dx += pc[s] * diff;
                d2x += pc[s] * pc[s];
                dy += pd[s] * diff;
                d2y += pd[s] * pd[s];
            }
        }
        if (d2x > 0.0f) fx -= dx / d2x;
        if (d2y > 0.0f) fy -= dy / d2y;
        if (fx > fy) { const float f = fx; fx = fy; fy = f; }
        if ((dx * dx < (1.0f / 64.0f)) && (dy * dy < (1.0f / 64.0f))) break;
    }
    *px = (fx < min_value) ? min_value : (fx > max_value) ? max_value : fx;
    *py = (fy < min_value) ? min_value : (fy > max_value) ? max_value : fy;
}

/* FindEndPointsBC4U (BC4BC5.cpp:186-238) */
static void find_endpoints_u(const float t[16], uint8_t* e0, uint8_t* e1)
{
    float bmax = t[0], bmin = t[0];
    for (int i = 0; i < 16; i++) {
        if (t[i] < bmin) bmin = t[i];
        else if (t[i] > bmax) bmax = t[i];
    }
    const int four_block = (0.0f == bmin || 1.0f == bmax);       /* "4 block-codec": six ramp steps + exact 0 and 1 */
    float fs, fe;
    if (!four_block) {
        optimize_alpha_unorm(&fs, &fe, t, 8);
        *e0 = (uint8_t)(fe * 255.0f);                            /* red_0 > red_1 selects the 8-value ramp */
        *e1 = (uint8_t)(fs * 255.0f);
    } else {
        optimize_alpha_unorm(&fs, &fe, t, 6);
        *e1 = (uint8_t)(fe * 255.0f);
        *e0 = (uint8_t)(fs * 255.0f);
    }
}

/* BC4_UNORM::DecodeFromIndex (BC4BC5.cpp:50-72) */
static float decode_from_index(uint8_t r0, uint8_t r1, int idx)
{
    if (idx == 0) return r0 / 255.0f;
    if (idx == 1) return r1 / 255.0f;
    const float f0 = r0 / 255.0f, f1 = r1 / 255.0f;
    if (r0 > r1) {
        idx -= 1;
        return (f0 * (float)(7 - idx) + f1 * (float)idx) / 7.0f;
    }
    if (idx == 6) return 0.0f;
    if (idx == 7) return 1.0f;
    idx -= 1;
    return (f0 * (float)(5 - idx) + f1 * (float)idx) / 5.0f;
}

/* one channel of one block: endpoints, then FindClosestUNORM (BC4BC5.cpp:314-337); 8 bytes out */
void oracle_bc4_block(const float t[16], uint8_t out[8])
{
    uint8_t r0, r1;
    find_endpoints_u(t, &r0, &r1);
    float grad[8];
    for (int i = 0; i < 8; i++) grad[i] = decode_from_index(r0, r1, i);
    uint64_t data = (uint64_t)r0 | ((uint64_t)r1 << 8);
    for (int i = 0; i < 16; i++) {
        int best = 0;
        float best_delta = 100000;
        for (int k = 0; k < 8; k++) {
            const float d = fabsf(grad[k] - t[i]);
            if (d < best_delta) { best = k; best_delta = d; }
        }
        data |= (uint64_t)best << (3 * i + 16);
    }
    for (int i = 0; i < 8; i++) out[i] = (uint8_t)(data >> (8 * i));
}

/* FindClosestUNORM (BC4BC5.cpp:314-337) as a function of the endpoint pair and ONE texel: out[v] = the index stored for a texel
 * of code v (value v * (1/255.f), header note 1), v = 0..255.  The whole domain is 256^3 cases; tests/test_bc45_index_table.py
 * walks it against the reference's own function and against the run-length form csrc/bc4_bc5.hip evaluates. */
void oracle_bc4_find_closest_row(int r0, int r1, uint8_t out[256])
{
    float grad[8];
    for (int i = 0; i < 8; i++) grad[i] = decode_from_index((uint8_t)r0, (uint8_t)r1, i);
    for (int v = 0; v < 256; v++) {
        const float t = (float)v * (1.0f / 255.0f);
        int best = 0;
        float best_delta = 100000;
        for (int k = 0; k < 8; k++) {
            const float d = fabsf(grad[k] - t);
            if (d < best_delta) { best = k; best_delta = d; }
        }
        out[v] = (uint8_t)best;
    }
}

/* DirectXTexCompress.cpp:105-168: load up to 4x4 texels of the RGBA8 surface, replicate into the missing columns and
 * rows with the source map {0,0,0,1}; channel -> float by XMLoadUByteN4's SSE path (header note 1). */
static void load_block(const oracle_surface* src, int bx, int by, int channel, float t[16])
{
    static const int usrc[4] = {0, 0, 0, 1};
    const int pw = (src->width - 4 * bx < 4) ? src->width - 4 * bx : 4;
    const int ph = (src->height - 4 * by < 4) ? src->height - 4 * by : 4;
    const float scale = 1.0f / 255.0f;
    for (int y = 0; y < ph; y++)
        for (int x = 0; x < pw; x++)
            t[y * 4 + x] = (float)src->ptr[(size_t)(4 * by + y) * src->stride + (size_t)(4 * bx + x) * 4 + channel] * scale;
    if (pw < 4)
        for (int y = 0; y < ph; y++)
            for (int x = pw; x < 4; x++) t[y * 4 + x] = t[y * 4 + usrc[x]];
    if (ph < 4)
        for (int y = ph; y < 4; y++)
            for (int x = 0; x < 4; x++) t[y * 4 + x] = t[usrc[y] * 4 + x];
}

static void compress(const oracle_surface* src, uint8_t* dst, int channels)
{
    const int nbx = (src->width + 3) / 4, nby = (src->height + 3) / 4;
    for (int by = 0; by < nby; by++)
        for (int bx = 0; bx < nbx; bx++)
            for (int c = 0; c < channels; c++) {
                float t[16];
                load_block(src, bx, by, c, t);
                oracle_bc4_block(t, dst + ((size_t)by * nbx + bx) * 8 * channels + 8 * c);
            }
}

/* any width, height >= 1; output ceil(w/4) x ceil(h/4) blocks, tightly packed (DirectXTex pitch rule) */
void oracle_CompressBlocksBC4(const oracle_surface* src, uint8_t* dst) { compress(src, dst, 1); }
void oracle_CompressBlocksBC5(const oracle_surface* src, uint8_t* dst) { compress(src, dst, 2); }

/* D3DXDecodeBC4U / BC5U (BC4BC5.cpp:373-385, 449-462): 16 floats per channel, raster order */
void oracle_decode_bc4_float(const uint8_t blk[8], float out[16])
{
    uint64_t data = 0;
    for (int i = 0; i < 8; i++) data |= (uint64_t)blk[i] << (8 * i);
    for (int i = 0; i < 16; i++) out[i] = decode_from_index(blk[0], blk[1], (int)((data >> (3 * i + 16)) & 7));
}
