/*
 * oracle/bc_common.h -- TEST INFRASTRUCTURE.  Helpers shared by the BC1/BC3,
 * BC7 and BC6H restatements: kernel.ispc:17-229.
 */
#ifndef ORACLE_BC_COMMON_H
#define ORACLE_BC_COMMON_H
#include "oracle.h"
#include "x86_math.h"

/* kernel.ispc:24-42 */
static inline void swap_ints(int32_t* u, int32_t* v, int n)
{ for (int i = 0; i < n; i++) { int32_t t = u[i]; u[i] = v[i]; v[i] = t; } }
static inline void swap_uints(uint32_t* u, uint32_t* v, int n)
{ for (int i = 0; i < n; i++) { uint32_t t = u[i]; u[i] = v[i]; v[i] = t; } }

static inline uint32_t load_u32(const uint8_t* p) { uint32_t v; memcpy(&v, p, 4); return v; }

/* kernel.ispc:105-117 -- RGB planes, alpha dropped */
static inline void load_block_interleaved(float block[48], const oracle_surface* src, int xx, int yy)
{
    for (int y = 0; y < 4; y++)
    for (int x = 0; x < 4; x++) {
        const uint8_t* row = src->ptr + (int64_t)(yy * 4 + y) * src->stride;
        uint32_t rgba = load_u32(row + 4 * (xx * 4 + x));
        block[16 * 0 + y * 4 + x] = (float)(int32_t)((rgba >> 0) & 255);
        block[16 * 1 + y * 4 + x] = (float)(int32_t)((rgba >> 8) & 255);
        block[16 * 2 + y * 4 + x] = (float)(int32_t)((rgba >> 16) & 255);
    }
}

/* kernel.ispc:119-132 */
static inline void load_block_interleaved_rgba(float block[64], const oracle_surface* src, int xx, int yy)
{
    for (int y = 0; y < 4; y++)
    for (int x = 0; x < 4; x++) {
        const uint8_t* row = src->ptr + (int64_t)(yy * 4 + y) * src->stride;
        uint32_t rgba = load_u32(row + 4 * (xx * 4 + x));
        block[16 * 0 + y * 4 + x] = (float)(int32_t)((rgba >> 0) & 255);
        block[16 * 1 + y * 4 + x] = (float)(int32_t)((rgba >> 8) & 255);
        block[16 * 2 + y * 4 + x] = (float)(int32_t)((rgba >> 16) & 255);
        block[16 * 3 + y * 4 + x] = (float)(int32_t)((rgba >> 24) & 255);
    }
}

/* kernel.ispc:134-151 -- raw half bit patterns as integers; plane 3 = 0.
 * The reference issues three overlapping u32 gathers at byte offsets +0/+2/+4
 * of each 8-byte texel and keeps the low 16 bits of each. */
static inline void load_block_interleaved_16bit(float block[64], const oracle_surface* src, int xx, int yy)
{
    for (int y = 0; y < 4; y++)
    for (int x = 0; x < 4; x++) {
        const uint8_t* row = src->ptr + (int64_t)(yy * 4 + y) * src->stride;
        const uint8_t* px = row + 8 * (xx * 4 + x);
        uint32_t xr = load_u32(px + 0), xg = load_u32(px + 2), xb = load_u32(px + 4);
        block[16 * 0 + y * 4 + x] = (float)(int32_t)(xr & 0xFFFF);
        block[16 * 1 + y * 4 + x] = (float)(int32_t)(xg & 0xFFFF);
        block[16 * 2 + y * 4 + x] = (float)(int32_t)(xb & 0xFFFF);
        block[16 * 3 + y * 4 + x] = 0.0f;
    }
}

/* kernel.ispc:153-160 -- tight output pitch (width/4)*data_size words */
static inline void store_data(uint8_t* dst, int width, int xx, int yy, const uint32_t* data, int data_size)
{
    for (int k = 0; k < data_size; k++) {
        /* dst is a byte array: row base = yy*width*data_size BYTES, then word index xx*data_size+k */
        uint8_t* p = dst + (int64_t)yy * width * data_size + 4 * (int64_t)(xx * data_size + k);
        memcpy(p, &data[k], 4);
    }
}

/* kernel.ispc:162-182 */
static inline void ssymm(float a[3], const float covar[6], const float b[3])
{
    a[0] = covar[0] * b[0] + covar[1] * b[1] + covar[2] * b[2];
    a[1] = covar[1] * b[0] + covar[3] * b[1] + covar[4] * b[2];
    a[2] = covar[2] * b[0] + covar[4] * b[1] + covar[5] * b[2];
}
static inline void ssymm3(float a[4], const float covar[10], const float b[4])
{
    a[0] = covar[0] * b[0] + covar[1] * b[1] + covar[2] * b[2];
    a[1] = covar[1] * b[0] + covar[4] * b[1] + covar[5] * b[2];
    a[2] = covar[2] * b[0] + covar[5] * b[1] + covar[7] * b[2];
}
static inline void ssymm4(float a[4], const float covar[10], const float b[4])
{
    a[0] = covar[0] * b[0] + covar[1] * b[1] + covar[2] * b[2] + covar[3] * b[3];
    a[1] = covar[1] * b[0] + covar[4] * b[1] + covar[5] * b[2] + covar[6] * b[3];
    a[2] = covar[2] * b[0] + covar[5] * b[1] + covar[7] * b[2] + covar[8] * b[3];
    a[3] = covar[3] * b[0] + covar[6] * b[1] + covar[8] * b[2] + covar[9] * b[3];
}

/* kernel.ispc:184-205 */
static inline void compute_axis3(float axis[3], const float covar[6], int powerIterations)
{
    float vec[3] = { 1, 1, 1 };
    for (int i = 0; i < powerIterations; i++) {
        ssymm(axis, covar, vec);
        for (int p = 0; p < 3; p++) vec[p] = axis[p];
        if (i % 2 == 1) {
            float norm_sq = 0;
            for (int p = 0; p < 3; p++) norm_sq += axis[p] * axis[p];
            float rnorm = ispc_rsqrt(norm_sq);
            for (int p = 0; p < 3; p++) vec[p] *= rnorm;
        }
    }
    for (int p = 0; p < 3; p++) axis[p] = vec[p];
}

/* kernel.ispc:207-229 */
static inline void compute_axis(float axis[4], const float covar[10], int powerIterations, int channels)
{
    float vec[4] = { 1, 1, 1, 1 };
    for (int i = 0; i < powerIterations; i++) {
        if (channels == 3) ssymm3(axis, covar, vec);
        if (channels == 4) ssymm4(axis, covar, vec);
        for (int p = 0; p < channels; p++) vec[p] = axis[p];
        if (i % 2 == 1) {
            float norm_sq = 0;
            for (int p = 0; p < channels; p++) norm_sq += axis[p] * axis[p];
            float rnorm = ispc_rsqrt(norm_sq);
            for (int p = 0; p < channels; p++) vec[p] *= rnorm;
        }
    }
    for (int p = 0; p < channels; p++) axis[p] = vec[p];
}

/* entry points of the other translation units */
void oracle_bc1_core(const float block[48], uint32_t data[2]);

#endif
