/*
 * itw_test_hooks.h -- entry points of libispc_texcomp_test.so ONLY (csrc/Makefile target `test`: the same sources compiled with
 * -DITW_TEST_HOOKS).  The product library, libispc_texcomp.so, does not export them (tests/test_abi_boundary.py checks `nm -D`): a drop-in
 * DLL carries no test code (VERDICT r04 / ADVICE r04).  tests/ load the test build for exactly these calls and the product for everything
 * else.  Replaces nothing in the reference.
 */
#ifndef ITW_TEST_HOOKS_H
#define ITW_TEST_HOOKS_H

#include "ispc_texcomp.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only what the headers declare is exported */

/* The pinned arithmetic on the GPU (csrc/x86_math.hpp): rcp / rsqrt / float->int of `n` floats resident in HBM. */
void itwTestRcp  (const float* d_in, float* d_out, int64_t n);
void itwTestRsqrt(const float* d_in, float* d_out, int64_t n);
void itwTestF2I  (const float* d_in, int32_t* d_out, int64_t n);

/* The bounded BC7 mode order's lower bound (csrc/bc7_exact.hpp two_subset_bound): for every 4x4 block of the device-resident RGBA8
 * surface, the bound of each of the 64 two-subset shapes, d_out[block * 64 + shape] (device memory, raster block order).  tests/ check it
 * against the oracle's error of every shape (it must never exceed one) and against its CPU restatement (oracle/bc7_bound.c). */
void itwTestBc7TwoSubsetBounds(const rgba_surface* d_src, float* d_out);

/* BC4 / BC5 (tests/test_gpu_parity_bc4_bc5.py).  The encoders evaluate FindClosestUNORM (BC4BC5.cpp:314-337) through a table
 * the current device builds once by running that search, as written, for all 65 536 endpoint pairs: per pair the 7 texel
 * codes where the chosen index changes and the 8 indices of the runs in between (csrc/bc4_bc5.hip).  Copies the table --
 * 65 536 entries x 4 words {256 - start of run 1..4, one byte each; the same for runs 5..7 with the number of runs in
 * the top byte; indices of runs 0..3, one byte each; indices of runs 4..7} -- to host memory.  Returns 0, or -1 on failure (error mode "return"). */
int itwTestBc45IndexTable(uint32_t* host_out);

/* Multi-GPU (tests/test_gpu_multigpu_cpp.py): the NEXT itwCompressImageMultiGPU[Ex] call of this process fails inside rank `rank` at `stage`
 * (1 = while preparing, before any transfer is posted; 2 = after its first half-band was posted; 3 = the rank stalls for
 * `stall_ms` before posting anything, which is what the watchdog is for).  One-shot; nothing in the environment can set it. */
void itwMultiGpuTestInjectFailure(int rank, int stage, int stall_ms);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
