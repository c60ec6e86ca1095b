/*
 * itw_amd.h -- MI355X-side extensions that sit next to the drop-in ABI of
 * ispc_texcomp.h.  Nothing here is needed by a caller that only wants the
 * reference behaviour; these entry points exist so a GPU-resident pipeline, a
 * benchmark or a multi-GPU driver can control placement and ordering.
 *
 * Pointer handling of CompressBlocks* (replaces nothing in the reference -- the
 * reference has no device; the device boundary sits exactly at
 * ispc_texcomp.cpp:417-435):
 *   src host, dst host     : staged H2D -> kernel -> D2H, synchronous (reference semantics)
 *   src device, dst device : kernel only, ASYNCHRONOUS on the calling thread's
 *                            stream (itwSetStream); caller synchronises
 *   mixed                  : the host side is staged, call returns synchronised
 * Device pointers must belong to the calling thread's current HIP device.
 */
#ifndef ITW_AMD_H
#define ITW_AMD_H

#include "ispc_texcomp.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Stream (hipStream_t as void*) used by this host thread's device-resident
 * calls.  Default: the NULL stream.  Thread-local. */
void  itwSetStream(void* hip_stream);
void* itwGetStream(void);

/* "gfx950 / <device name> / <CU count> CUs" of the current device; static storage per thread. */
const char* itwDeviceInfo(void);

/* Library build identification: arithmetic model and arch, e.g.
 * "itw-amd 0.1 gfx950 arith=x86-lut-nr contract=off". */
const char* itwVersion(void);

/* Row-band rule used to shard one surface over `parts` workers (GPUs or host
 * threads) -- the 4-row-aligned banding of win32Threads.cpp:217-231 restated on
 * block rows: part p owns block rows [R*p/parts, R*(p+1)/parts), R = height/4.
 * Writes first texel row and texel-row count; returns the byte offset of the
 * band in the tightly packed output (bytes_per_block = 8 or 16). */
int64_t itwBandForPart(int32_t width, int32_t height, int32_t bytes_per_block,
                       int32_t part, int32_t parts, int32_t* first_row, int32_t* row_count);

/* Device-side self test hooks (used by tests/ to prove the pinned arithmetic on
 * the GPU): evaluate rcp / rsqrt / float->int of `n` floats resident in HBM. */
void itwTestRcp  (const float* d_in, float* d_out, int64_t n);
void itwTestRsqrt(const float* d_in, float* d_out, int64_t n);
void itwTestF2I  (const float* d_in, int32_t* d_out, int64_t n);

#ifdef __cplusplus
}
#endif
#endif
