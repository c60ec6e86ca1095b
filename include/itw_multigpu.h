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
 */
#ifndef ITW_MULTIGPU_H
#define ITW_MULTIGPU_H

#include "itw_dispatch.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Ranks a call with ranks = 0 uses: ITW_MULTIGPU_RANKS if set, else the number of visible devices.  Rank r runs on
 * device r % device_count, so more ranks than devices is legal (how the 8-way path is exercised on a 1-GPU box). */
int itwMultiGpuRanks(void);

/* "rccl" or "peer": what the last call on this process used for device-resident gathers (static storage). */
const char* itwMultiGpuTransport(void);

/* Directed device pairs for which the rank threads enabled peer access so far (hipDeviceEnablePeerAccess): scatter and
 * peer-copy gather then travel over xGMI instead of bouncing through host memory.  0 on a one-GPU box. */
int itwMultiGpuPeerLinks(void);

/* Encode `input` with `cmpFunc` (a CompressImage* trampoline, win32Threads.h:58-80) across `ranks` ranks and leave the
 * whole block stream in `output`.  Pointers: host or device, as for CompressBlocks*.  Synchronous.  Returns false only in
 * error mode "return" (itwSetErrorMode) when some rank failed; itwLastError() then holds the message. */
bool itwCompressImageMultiGPU(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format, int ranks);

#ifdef __cplusplus
}
#endif
#endif
