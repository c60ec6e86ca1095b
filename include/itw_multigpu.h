/*
 * itw_multigpu.h -- one surface, all GPUs of the node, one process, host code in C++.
 *
 * The reference shards a surface by rows twice -- 0x40000-pixel slices (IntelPlugin.cpp:851-879) and one 4-row-aligned
 * band per pool thread, each band an independent CompressBlocks* call writing at dst + row0*(width/4)*bytes_per_block
 * (win32Threads.cpp:211-249).  Blocks never interact, so the same rule shards across GPUs: rank r of R encodes block
 * rows [B*r/R, B*(r+1)/R) (itwBandForPart).  The only exchange is the gather of the compressed bands to whoever owns
 * `output`:
 *   output in host memory          every GPU downloads its own band over its own PCIe link;
 *   output resident on GPU g       the other GPUs send their bands to g over xGMI -- RCCL grouped ncclSend / ncclRecv on
 *                                  communicators from ncclCommInitAll (librccl is loaded on first use; the library does
 *                                  not link it), or hipMemcpyPeerAsync when RCCL is unavailable, when several ranks share
 *                                  a device, or when ITW_MULTIGPU_TRANSPORT=peer.
 * Input texels: host memory is uploaded band by band by the GPU that encodes the band; a surface resident on one GPU is
 * scattered to the others with peer copies (the owner's band is encoded in place).
 * Each rank cuts its band in two: the gather of the first half runs on a second stream while the second half encodes.
 * Strides are signed like the reference's (bottom-up surfaces are staged row by row, as CompressBlocks* does).
 *
 * Failures: the ranks allocate everything first and post transfers only if all of them are ready; a failure after that
 * aborts the RCCL communicators so no rank keeps waiting for a peer that will not send (rebuilt on the next call).  The
 * call then fails as a whole: abort() with a diagnostic, or `false` + itwLastError() under ITW_ON_ERROR_RETURN.
 * A watchdog covers the one thing a failure model cannot: a rank that neither fails nor proceeds (itwCompressImageMultiGPUEx).
 */
#ifndef ITW_MULTIGPU_H
#define ITW_MULTIGPU_H

#include <stdint.h>
#include "itw_dispatch.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only what the headers declare is exported */

/* Ranks a call with ranks = 0 uses: ITW_MULTIGPU_RANKS if set, else the number of visible devices.  Rank r runs on
 * device r % device_count, so more ranks than devices is legal (how the 8-way path is exercised on a 1-GPU box). */
int itwMultiGpuRanks(void);

/* PARTITION (round 5).  K = sub-bands per rank.  K = 1 is the reference's rule (win32Threads.cpp:217-231 restated on block rows): rank r owns
 * the contiguous band itwBandForPart(r, ranks), cut in two halves for overlap.  K > 1: the surface is cut into K * ranks sub-bands
 * (itwBandForPart(j, K * ranks)) and sub-band j belongs to rank j % ranks -- every rank gets K stripes spread over the surface.  Since the
 * bounded BC7 mode order a band's encode time depends on its content (1.4x between noise and a photograph), so contiguous bands of a
 * mixed image leave the ranks unequally loaded; interleaved ones do not.  The output layout does not change: a rank's gather is K
 * contiguous runs.  itwMultiGpuSetInterleave sets the requested K for the process (1..8; default 4, env ITW_MULTIGPU_INTERLEAVE); a call
 * uses it where every sub-band still has 16 block rows, else K = 1: itwMultiGpuPieces(height, ranks, keep_partial_blocks) returns the K a
 * call with that geometry uses. */
void itwMultiGpuSetInterleave(int k);
int  itwMultiGpuPieces(int32_t height, int ranks, int keep_partial_blocks);

/* "rccl" or "peer": what the last call on this process used for device-resident gathers (static storage). */
const char* itwMultiGpuTransport(void);

/* Directed device pairs for which the rank threads enabled peer access so far (hipDeviceEnablePeerAccess): scatter and
 * peer-copy gather then travel over xGMI instead of bouncing through host memory.  0 on a one-GPU box. */
int itwMultiGpuPeerLinks(void);

/* Encode `input` with `cmpFunc` (a CompressImage* trampoline, win32Threads.h:58-80) across `ranks` ranks and leave the
 * whole block stream in `output`.  Pointers: host or device, as for CompressBlocks*.  Synchronous.  Returns false only in
 * error mode "return" (itwSetErrorMode) when some rank failed; itwLastError() then holds the message. */
bool itwCompressImageMultiGPU(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks);

/* What one call did, per rank (device-side durations from HIP events on the rank's own streams, summed over its pieces:
 * two half-bands, or K sub-bands) and as a whole.  `transport` = "rccl" | "peer" | "host" (output in host memory: every GPU downloads its own
 * band); `transport_note` says why RCCL was not used when it was not.  `rccl_ranks` = size of the communicator clique
 * the gather ran on (0 unless transport is "rccl"). */
typedef struct itw_multigpu_rank_stats {
    int32_t rank, device;
    int32_t block_row0, block_rows;      /* first block row of the rank's first piece; block rows of all its pieces together */
    float   upload_ms;                   /* host->GPU or owner GPU->GPU copies of the band's texels (0 when encoded in place) */
    float   encode_ms;                   /* the CompressBlocks* launches */
    float   gather_ms;                   /* D2H / ncclSend / peer copy of the band's blocks; on the owner rank: its grouped ncclRecv */
    float   span_ms;                     /* first device-side event to last: upload, encode and gather overlap inside it */
} itw_multigpu_rank_stats;

typedef struct itw_multigpu_stats {
    int32_t ranks, devices, peer_links, rccl_ranks;
    int32_t watchdog_fired;              /* 1 = a rank did not finish posting its work in time and the call was aborted */
    int32_t resident_bands;              /* 1 = the texels were already on the ranks' devices (no scatter) */
    float   wall_ms;                     /* host wall clock of the whole call */
    float   posted_ms;                   /* host wall clock until every rank had posted all its work (launch + enqueue cost) */
    char    transport[8];
    char    transport_note[96];
    itw_multigpu_rank_stats rank[64];
    /* fields added after round 4 are appended, so the offsets above never move: */
    int32_t interleave;                  /* K the call used (1 = one contiguous band per rank) */
} itw_multigpu_stats;

/* The same call with two optional extras.
 *   resident_bands  NULL, or an array of `ranks` surfaces: band r of the image (block rows itwBandForPart(r, ranks)), already resident on
 *                   the device rank r runs on (device r % device_count) -- the tile-sharded input of a pipeline that produced the texels
 *                   where they are encoded.  No scatter happens; `input` then only carries width / height (its ptr may be NULL); ranks
 *                   must be given explicitly (> 0).  A call with resident bands always uses ONE band per rank (K = 1), whatever
 *                   itwMultiGpuSetInterleave says: the array's length is the caller's word, and this entry point's word is `ranks`.
 *   stats           NULL, or where to leave the call's account (filled on failure too, as far as the call got).
 * Watchdog: a rank that has not posted all its work (launches, copies, ncclSend / ncclGroupEnd) within
 * ITW_MULTIGPU_POST_TIMEOUT_S (default 30) -- the first RCCL call of a process sets up its peer connections inside those
 * calls -- or a call that has not finished within ITW_MULTIGPU_TIMEOUT_S (default 600) is aborted through the same path
 * as a failing rank (ncclCommAbort on every communicator); the call fails with "watchdog" in the message. */
bool itwCompressImageMultiGPUEx(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks,
                                const rgba_surface* resident_bands, itw_multigpu_stats* stats);

/* Resident input cut into K interleaved sub-bands per rank: `resident_bands` holds n_resident_bands = K * ranks surfaces (K = 1 .. 8),
 * sub-band j = block rows itwBandForPart(j, K * ranks) resident on the device of rank j % ranks.  K is stated by THIS call through the
 * array's length -- it is validated (a multiple of `ranks`, at most 8 per rank, no more sub-bands than block rows; `ranks` itself at most
 * 64 and at most the number of block rows) and never taken from process-wide state.  Everything else as itwCompressImageMultiGPUEx. */
bool itwCompressImageMultiGPUBands(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks,
                                   const rgba_surface* resident_bands, int n_resident_bands, itw_multigpu_stats* stats);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
