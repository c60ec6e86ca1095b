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
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only what the headers declare is exported */

/* Stream (hipStream_t as void*) used by this host thread's device-resident
 * calls.  Default: the NULL stream.  Thread-local and sticky: the handle must stay valid for as long as it is the
 * thread's current stream (reset with itwSetStream(NULL) before destroying it).  The library keeps no other reference
 * to caller streams: cross-stream ordering of its per-thread BC7 workspace uses library-owned events. */
void  itwSetStream(void* hip_stream);
void* itwGetStream(void);

/* Failure handling.  The reference ABI is void and cannot fail (ispc_texcomp.h:104-107); this library needs a GPU
 * and has, on purpose, no CPU path behind it.
 *   itwAvailable()   1 if the calling thread's current HIP device exists and is a gfx950 (the only code object in the
 *                    library); 0 otherwise.  Never aborts, never prints: a host decides BEFORE calling CompressBlocks*.
 *   itwSetErrorMode  process-wide.  ITW_ON_ERROR_ABORT (default; also env ITW_ON_ERROR=abort): a HIP failure (no
 *                    device, out of memory, invalid pointer) prints a diagnostic and abort()s -- loud, never a
 *                    silently wrong texture.  ITW_ON_ERROR_RETURN (env ITW_ON_ERROR=return): the failing call returns
 *                    (destination contents undefined), the message is kept for the calling host thread, the
 *                    bool-returning dispatch entry points (CompressImageMT/ST, itwCompressImageSliced) return false.
 *   itwLastError()   message of the last failed call on this host thread, NULL if the last ABI call succeeded
 *                    (every CompressBlocks* call clears it on entry).  Static storage per thread. */
enum { ITW_ON_ERROR_ABORT = 0, ITW_ON_ERROR_RETURN = 1 };
int         itwAvailable(void);
void        itwSetErrorMode(int mode);
const char* itwLastError(void);
void        itwClearError(void);

/* "gfx950 / <device name> / <CU count> CUs" of the current device; static storage per thread. */
const char* itwDeviceInfo(void);

/* Tuning / test knob: how a BC7 call is laid out on the GPU.  ITW_BC7_PATH_AUTO (default; env ITW_BC7_PATH=deep|wide
 * presets it) picks by call size: DEEP = one lane per block, one launch pair per mode family (fills the chip on whole
 * surfaces); WIDE = every family's scan split over several waves, winners joined by an ordered argmin (calls too small
 * to fill the chip: the plugin's 0x40000-pixel slices, IntelPlugin.cpp:851, and the per-thread bands of
 * win32Threads.cpp:217).  Both produce the same bytes.  The same switch governs the BC6H slow profiles (one kernel / split
 * two-region scan; env ITW_BC6H_PATH). */
enum { ITW_BC7_PATH_AUTO = 0, ITW_BC7_PATH_DEEP = 1, ITW_BC7_PATH_WIDE = 2 };
void itwSetBc7Path(int path);

/* Tuning / test knob of the BC7 `slow` profile's mode order on whole surfaces (csrc/bc7.hip): the library can run modes 1/3 last and only
 * for the blocks an exact lower bound cannot exclude ("bounded order"), which pays on content where few blocks need them and costs a few
 * percent where nearly all do.  A pilot decides per call, on the device: behind the first band's {0,2} scan -- common to both orders -- an
 * estimate kernel (bc7_pilot_estimate) evaluates the bound against the scan's winners on 1/16 of the surface (every eighth chunk of that
 * band) and leaves a device word; both continuations of each band are enqueued behind it, each launch gated on that word, and the one not
 * chosen returns at once (no host round trip).  `percent` = the share of the sampled blocks the estimate may list for modes 1/3 for the
 * call to take the bounded order (default 90; env ITW_BC7_PILOT_THR presets it).  0 = the reference's order unless the estimate lists no
 * block at all (then the bounded order, which skips modes 1/3 everywhere), 100 = always the bounded order, -1 = no pilot (the whole call in
 * the bounded order, ungated), any value below -1 = back to the preset (the environment's, else 90).  The emitted bytes are the same
 * whatever the value. */
void itwSetBc7Pilot(int percent);

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

/* The same rule for the formats that keep partial blocks (BC4 / BC5, itw_bc45.h): R = ceil(height/4), ceil(width/4) blocks
 * per row, and the band that holds the last block row ends at `height` (keep_partial_blocks = 0: identical to the above). */
int64_t itwBandForPartEx(int32_t width, int32_t height, int32_t bytes_per_block, int32_t part, int32_t parts,
                         int32_t keep_partial_blocks, int32_t* first_row, int32_t* row_count);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
