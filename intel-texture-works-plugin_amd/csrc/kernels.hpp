// kernels.hpp -- host-visible launchers of the gfx950 encoder kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ispc_texcomp.h"

namespace itw {

// All launchers: `src`/`dst` are DEVICE pointers, rows `stride` bytes apart,
// output tightly packed in raster block order; asynchronous on `st`.
void launch_bc1 (const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st);
void launch_bc3 (const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st);
// BC4_UNORM / BC5_UNORM of the R (and G) channel of an RGBA8 surface, DirectXTex's encoder (bc4_bc5.hip).  Any
// width/height >= 1: ceil(width/4) x ceil(height/4) blocks, partial blocks replicated by DirectXTex's rule.
void launch_bc4 (const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st);
void launch_bc5 (const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st);
void warmup_bc45();                                               // builds the current device's FindClosestUNORM run table now (else: first call)
void copy_bc45_index_table(uint32_t* host_out, hipStream_t st);   // test hook: the FindClosestUNORM run table of the current device
// BC7 runs as up to seven kernels (search + finish per multi-subset mode family, one for modes 4/5/6) that hand
// "best error so far" and the search winners to each other through
// `workspace`: device memory, bc7_workspace_bytes(width, height) bytes, 16 B aligned, contents irrelevant on entry.
// `settings` (optional): size for what a call with these settings can touch -- the lists, band regions and compact texel copy belong to the
// alpha-first and bounded mode orders (36 B per block without them, ~124 with); nullptr sizes for any settings.
size_t bc7_workspace_bytes(int width, int height, int64_t wide_max_blocks = 0, const bc7_enc_settings* settings = nullptr);   // wide_max_blocks as in Bc7Aux
// BC7 launch shape: 0 = by call size (default), 1 = deep (one lane per block, a launch pair per mode family: fills the
// chip on whole surfaces), 2 = wide (split scans + ordered argmin: small calls).  Same blocks either way.
void set_bc7_path(int path);
bool bc7_scans_every_shape(const bc7_enc_settings& s);  // the settings take the bounded mode order (modes 1/3 -- and 7 -- scan all 64 shapes: `slow`, `alpha_slow`): about twice the work per block
bool bc7_has_order_verdict(const bc7_enc_settings& s);   // the settings run the bounded order of the RGB profiles (the one with a pilot's estimate)
bool bc7_staged_bands_ok();      // the staged runs of a host-pointer call may run as overlapped deep bands (not when the wide shape is forced / ITW_STAGED_BANDS=0)
// The pilot of the bounded BC7 mode order (bc7.hip, launch_bc7): percent of the blocks the pilot looks at that its estimate may list for modes 1/3 for the rest of
// the surface to take the bounded order; -1 = no pilot, the whole call in the bounded order.  Same blocks whatever the value.
void set_bc7_pilot(int percent);
// `aux` (optional): a second stream of the same device plus two events the launcher may use to run independent parts of a
// small call side by side; everything is joined back into `st` before the launcher returns.
// `mid` (may be null: no pilot): a third event, for the pilot of the bounded mode order (bc7.hip).  `single`: the call is one band of a
// larger job whose bands the CALLER overlaps on two streams (the staged runs of a host-pointer call, abi.hip): deep shape whatever the
// size, everything on `st`, no pilot, no inner bands.
// `verdict` (optional, `single` calls): where a call that runs the bounded order leaves the pilot's estimate for the HOST -- {blocks some
// two-subset shape can still improve, blocks looked at}, written by the estimate kernel's last workgroup straight into PINNED HOST memory
// (`host_counts_dev` = the device-side address of the caller's two pinned words, `host_counts` = their host address), complete once `event`
// (recorded behind that kernel) has fired: the host reads two words, no copy on any stream.  The caller of the first staged run polls it
// under the upload of the next run and picks the launch shape of the remaining runs (abi.hip).
struct Bc7Verdict { hipEvent_t event; volatile int32_t* host_counts; bool valid; int32_t* host_counts_dev; };
// `probe` (with `single` and `verdict`): only the pilot's estimate is computed -- the {0,2} scan of every eighth chunk and the count -- nothing
// is encoded (a staged call whose runs take the wide shape keeps its verdict fresh this way).
struct Bc7Aux { hipStream_t stream; hipEvent_t fork, join; int64_t wide_max_blocks; hipEvent_t mid; bool single; Bc7Verdict* verdict; bool probe; };   // wide_max_blocks: 0 = the library default
void launch_bc7 (const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst,
                 const bc7_enc_settings& s, float* workspace, hipStream_t st, const Bc7Aux* aux = nullptr);
// test hook: bc7_exact.hpp's two_subset_bound of all 64 two-subset shapes of every block, out[block * 64 + shape] (device memory)
void launch_bc7_test_bounds(const uint8_t* src, int64_t stride, int width, int height, float* out, hipStream_t st);
// `workspace`: device memory of bc6h_workspace_bytes(width, height, s) bytes (0 for most calls: only small calls of the slow
// profiles, which take the wide shape, need any); nullptr keeps the one-kernel path.
size_t bc6h_workspace_bytes(int width, int height, const bc6h_enc_settings& s);
void launch_bc6h(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst,
                 const bc6h_enc_settings& s, hipStream_t st, void* workspace = nullptr);
void set_bc6h_path(int path);         // 0 by size, 1 one kernel, 2 wide (tests / probes)

} // namespace itw
