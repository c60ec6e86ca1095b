/*
 * oracle/oracle.h -- TEST INFRASTRUCTURE: public face of the CPU parity oracle.
 *
 * A scalar, one-lane-at-a-time C restatement of the reference's SPMD encoders
 * (/root/reference/IntelCompressionPlugin/kernel.ispc:17-3139) under the pinned
 * arithmetic of x86_math.h.  ISPC lanes never communicate in these kernels
 * (kernel.ispc:598-614, 2030-2037, 3132-3139: one 4x4 block per program
 * instance), so a scalar loop over blocks is an exact model of the gang.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * this library.  The product (libispc_texcomp.so) never links or calls it.
 *
 * PARITY: the reference ships no golden outputs and its ispc build cannot be made
 * here; pinned instead by the reference's kernel.ispc compiled as ONE scalar program
 * instance (oracle/ref_build/ispc_as_cpp/, oracle/_ref/libispc_texcomp_ref_full.so):
 * byte-identical to this restatement on every golden input, preset, random settings
 * struct and both arithmetic-model switches (tests/test_reference_kernel_source.py).
 * Assumed, not pinned: the ispc compiler / stdlib semantics (x86_math.h S2-S6).
 *
 * Symbols carry an oracle_ prefix so both libraries can live in one process.
 */
#ifndef ORACLE_H
#define ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* layout twins of ispc_texcomp.h:19-50 (kept separate on purpose: the oracle
 * does not include product headers) */
typedef struct { uint8_t* ptr; int32_t width, height, stride; } oracle_surface;

typedef struct {
    uint8_t mode_selection[4];
    int32_t refineIterations[8];
    uint8_t skip_mode2;
    int32_t fastSkipTreshold_mode1, fastSkipTreshold_mode3, fastSkipTreshold_mode7;
    int32_t mode45_channel0;
    int32_t refineIterations_channel;
    int32_t channels;
} oracle_bc7_settings;

typedef struct {
    uint8_t slow_mode, fast_mode;
    int32_t refineIterations_1p, refineIterations_2p, fastSkipTreshold;
} oracle_bc6h_settings;

void oracle_CompressBlocksBC1 (const oracle_surface* src, uint8_t* dst);
void oracle_CompressBlocksBC3 (const oracle_surface* src, uint8_t* dst);
void oracle_CompressBlocksBC7 (const oracle_surface* src, uint8_t* dst, const oracle_bc7_settings* s);
void oracle_CompressBlocksBC6H(const oracle_surface* src, uint8_t* dst, const oracle_bc6h_settings* s);

/* name = "ultrafast" | "veryfast" | "fast" | "basic" | "slow" | "alpha_*";
 * returns 0 on success.  Fields a reference profile leaves unwritten stay as
 * they were in *s (ispc_texcomp.cpp:20-189 never writes refineIterations[7]). */
int oracle_GetProfile_bc7 (const char* name, oracle_bc7_settings* s);
int oracle_GetProfile_bc6h(const char* name, oracle_bc6h_settings* s);

/* single-block entry points for unit tests: block = planar floats as built by
 * load_block_interleaved* (kernel.ispc:105-151) */
void oracle_bc1_block(const float block[48], uint32_t data[2]);
void oracle_bc3_alpha_block(const float alpha[16], uint32_t data[2]);
void oracle_bc7_block(const float block[64], const oracle_bc7_settings* s, uint32_t data[4], float* best_err);
float oracle_bc7_two_subset_bound(const float block[64], int shape);   /* bc7_bound.c: restatement of the product's bounded-order bound */
float oracle_bc7_one_line_bound(const float block[64]);                /* bc7_bound.c: ... and of the bound that lets mode 6 be skipped (RGB profiles) */
void oracle_bc7_part_fast_errors(const float block[64], int mode, float err[64], int32_t key[64]);   /* study / test hook */
void oracle_bc6h_block(const float block[64], const oracle_bc6h_settings* s, uint32_t data[4], float* best_err);

/* arithmetic primitives, exported for the LUT / NR self tests */
float   oracle_rcp(float v);
float   oracle_rsqrt(float v);
float   oracle_rcpps(float v);
float   oracle_rsqrtps(float v);
int32_t oracle_f2i(float v);

/* from-spec decoders (independent check that emitted blocks are valid):
 * out = 16 texels, RGBA8 for BC1/BC3/BC7, 3 x uint16 half bit patterns for BC6H */
void oracle_decode_bc1 (const uint8_t blk[8],  uint8_t out_rgba[64]);
void oracle_decode_bc3 (const uint8_t blk[16], uint8_t out_rgba[64]);
int  oracle_decode_bc7 (const uint8_t blk[16], uint8_t out_rgba[64]);   /* returns mode, -1 if reserved */
int  oracle_decode_bc6h(const uint8_t blk[16], uint16_t out_rgb[48]);  /* unsigned; returns mode 0..13 (kernel.ispc numbering), -1 if reserved */

/* BC4_UNORM / BC5_UNORM: the DirectXTex encoder the plugin uses for these two formats (bc4_bc5.c).  Source = RGBA8
 * surface (R, or R and G, are encoded); any width/height >= 1, ceil(w/4) x ceil(h/4) blocks out. */
void oracle_CompressBlocksBC4(const oracle_surface* src, uint8_t* dst);
void oracle_CompressBlocksBC5(const oracle_surface* src, uint8_t* dst);
void oracle_bc4_block(const float texels[16], uint8_t out[8]);
void oracle_bc4_find_closest_row(int r0, int r1, uint8_t out[256]);   /* FindClosestUNORM for texel codes 0..255 */
void oracle_decode_bc4_float(const uint8_t blk[8], float out[16]);   /* DirectXTex's float decode (BC4BC5.cpp:50-72) */
void oracle_decode_bc4_rgba8(const uint8_t blk[8], uint8_t out_rgba[64]);   /* from the format definition, (R,0,0,255) */
void oracle_decode_bc5_rgba8(const uint8_t blk[16], uint8_t out_rgba[64]);  /* (R,G,0,255) */

#ifdef __cplusplus
}
#endif
#endif
