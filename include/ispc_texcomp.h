/*
 * ispc_texcomp.h -- the drop-in C ABI of the MI355X-native BCn encoder.
 *
 * This header declares exactly the entry points and POD structs that the
 * reference's L1 boundary exposes for the BC1/BC3/BC7/BC6H path
 *   /root/reference/3rdParty/Intel/Source/ispc_texcomp.h:19-50   (structs)
 *   /root/reference/3rdParty/Intel/Source/ispc_texcomp.h:67-85   (GetProfile_*)
 *   /root/reference/3rdParty/Intel/Source/ispc_texcomp.h:104-107 (CompressBlocks*)
 * so that its callers (win32Threads.cpp:289-329 trampolines, IntelPlugin.cpp:816)
 * link against libispc_texcomp.so unchanged.  Layouts are asserted at the end.
 *
 * Not provided (outside the accelerated path):
 * CompressBlocksETC1 / GetProfile_etc_slow / the ASTC declarations.
 *
 * Contract (same as the reference, ispc_texcomp.h:93-102, corrected where the
 * reference comment is wrong):
 *   - src->ptr: top-left texel; rows are src->stride BYTES apart.
 *       BC1/BC3/BC7: 8-bit RGBA, R in the lowest byte.  BC6H: RGBA16F bit patterns.
 *   - width/height are in texels and should be multiples of 4; partial blocks are
 *     dropped (kernel.ispc:600-601 semantics: height/4 x width/4 blocks).
 *   - dst: caller-allocated, blocks in raster order, tightly packed,
 *       8 bytes/block for BC1, 16 bytes/block for BC3/BC7/BC6H
 *     (kernel.ispc:582,595,2027,3129; the "4/8 bytes" remark in the reference
 *     header is off by 2x).
 *   - void return, no errno.  There is no CPU fallback: a HIP failure (no device, out of
 *     memory, bad pointer) either prints a diagnostic and abort()s (the default: loud,
 *     never a silently wrong texture) or, after itwSetErrorMode(ITW_ON_ERROR_RETURN)
 *     (include/itw_amd.h), makes the call return with the message kept per host thread in
 *     itwLastError(); itwAvailable() lets a host decide before it calls.
 *   - re-entrant; may be called concurrently from many host threads on disjoint
 *     row bands (win32Threads.cpp:211-274 does exactly that).
 *
 * MI355X extension (does not change the ABI): src->ptr and dst may each be a
 * host pointer or a HIP device pointer; see include/itw_amd.h.
 */
#ifndef ISPC_TEXCOMP_H
#define ISPC_TEXCOMP_H

#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only what the headers declare is exported */

struct rgba_surface
{
    uint8_t* ptr;
    int32_t  width;
    int32_t  height;
    int32_t  stride;   /* bytes */
};

struct bc7_enc_settings
{
    bool mode_selection[4];        /* [0] modes 0+2, [1] modes 1+3+7, [2] modes 4+5, [3] mode 6 */
    int  refineIterations[8];      /* per BC7 mode */

    bool skip_mode2;
    int  fastSkipTreshold_mode1;   /* how many PCA-ranked partitions each mode tries */
    int  fastSkipTreshold_mode3;
    int  fastSkipTreshold_mode7;

    int  mode45_channel0;          /* first rotation candidate for modes 4/5 */
    int  refineIterations_channel;

    int  channels;                 /* 3 = RGB profile (alpha ignored), 4 = RGBA */
};

struct bc6h_enc_settings
{
    bool slow_mode;
    bool fast_mode;
    int  refineIterations_1p;
    int  refineIterations_2p;
    int  fastSkipTreshold;
};

#ifndef __cplusplus
typedef struct rgba_surface      rgba_surface;
typedef struct bc7_enc_settings  bc7_enc_settings;
typedef struct bc6h_enc_settings bc6h_enc_settings;
#endif

/* BC7, opaque sources (alpha ignored) -- ispc_texcomp.cpp:20-189 */
void GetProfile_ultrafast(bc7_enc_settings* settings);
void GetProfile_veryfast (bc7_enc_settings* settings);
void GetProfile_fast     (bc7_enc_settings* settings);
void GetProfile_basic    (bc7_enc_settings* settings);
void GetProfile_slow     (bc7_enc_settings* settings);

/* BC7, sources with alpha -- ispc_texcomp.cpp:191-365 */
void GetProfile_alpha_ultrafast(bc7_enc_settings* settings);
void GetProfile_alpha_veryfast (bc7_enc_settings* settings);
void GetProfile_alpha_fast     (bc7_enc_settings* settings);
void GetProfile_alpha_basic    (bc7_enc_settings* settings);
void GetProfile_alpha_slow     (bc7_enc_settings* settings);

/* BC6H (unsigned half float RGB) -- ispc_texcomp.cpp:367-410 */
void GetProfile_bc6h_veryfast(bc6h_enc_settings* settings);
void GetProfile_bc6h_fast    (bc6h_enc_settings* settings);
void GetProfile_bc6h_basic   (bc6h_enc_settings* settings);
void GetProfile_bc6h_slow    (bc6h_enc_settings* settings);
void GetProfile_bc6h_veryslow(bc6h_enc_settings* settings);

/* the hot path -- ispc_texcomp.cpp:417-435 -> kernel.ispc:598,607,2030,3132 */
void CompressBlocksBC1 (const rgba_surface* src, uint8_t* dst);
void CompressBlocksBC3 (const rgba_surface* src, uint8_t* dst);
void CompressBlocksBC6H(const rgba_surface* src, uint8_t* dst, bc6h_enc_settings* settings);
void CompressBlocksBC7 (const rgba_surface* src, uint8_t* dst, bc7_enc_settings* settings);

#pragma GCC visibility pop
#ifdef __cplusplus
}
static_assert(sizeof(rgba_surface) == 24,      "rgba_surface layout (x86-64)");
static_assert(sizeof(bc7_enc_settings) == 64,  "bc7_enc_settings layout");
static_assert(sizeof(bc6h_enc_settings) == 16, "bc6h_enc_settings layout");
#endif

#endif /* ISPC_TEXCOMP_H */
