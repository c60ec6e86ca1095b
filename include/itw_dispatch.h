/*
 * itw_dispatch.h -- portable restatement of the reference's threading / dispatch layer
 * (3rdParty/Intel/Source/win32Threads.h:24-80, win32Threads.cpp:98-329): the immediate caller of the
 * CompressBlocks* C ABI.  Same entry points, same argument meaning; Win32 types replaced by portable ones
 * (BYTE -> uint8_t, DXGI_FORMAT -> int carrying the DXGI_FORMAT_* value) and C linkage, so any host language can
 * bind it.
 *
 * What changes underneath: the reference keeps a pool of up to 64 Win32 threads and hands each a 4-row-aligned band
 * of the surface (win32Threads.cpp:211-249).  Here a "worker" is a GPU: CompressImageMT cuts the surface with the
 * same band rule over GetProcessorCount() = number of visible MI355X devices and runs one band per device from a
 * persistent pool of host threads (one per device), so a single-GPU box makes ONE whole-surface call instead of
 * dozens of few-thousand-pixel calls, and an 8-GPU node encodes 8 bands concurrently in one process.
 */
#ifndef ITW_DISPATCH_H
#define ITW_DISPATCH_H

#include "ispc_texcomp.h"

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default)   /* the library is built with -fvisibility=hidden: only what the headers declare is exported */

/* win32Threads.h:24 */
typedef void (CompressionFunc)(const rgba_surface* input, uint8_t* output);

/* the DXGI_FORMAT values the dispatch layer understands (dxgiformat.h; win32Threads.cpp:192-209) */
enum {
    ITW_DXGI_FORMAT_BC1_UNORM = 71, ITW_DXGI_FORMAT_BC1_UNORM_SRGB = 72,
    ITW_DXGI_FORMAT_BC3_UNORM = 77, ITW_DXGI_FORMAT_BC3_UNORM_SRGB = 78,
    ITW_DXGI_FORMAT_BC4_UNORM = 80, ITW_DXGI_FORMAT_BC5_UNORM = 83,       /* the DirectXTex formats, itw_bc45.h */
    ITW_DXGI_FORMAT_BC6H_UF16 = 95, ITW_DXGI_FORMAT_BC6H_SF16 = 96,
    ITW_DXGI_FORMAT_BC7_UNORM = 98, ITW_DXGI_FORMAT_BC7_UNORM_SRGB = 99
};

/* win32Threads.h:52-55 / win32Threads.cpp:98-190.  GetProcessorCount(): number of workers = visible GPUs (>= 1), or the
 * value of the environment variable ITW_WORKERS when set (extra workers share the devices round-robin).
 * InitWin32Threads() starts the pool (idempotent); DestroyThreads() joins it.  CompressImageMT initialises lazily. */
int  GetProcessorCount(void);
void InitWin32Threads(void);
void DestroyThreads(void);

/* win32Threads.cpp:192-209: 8 for BC1 (and anything unknown, so BC4 too), 16 for BC3 / BC7 / BC6H -- and for BC5, which
 * the reference's switch does not list because BC5 never reaches it there */
int  GetBytesPerBlock(int dxgi_format);

/* win32Threads.cpp:211-249, 277-282.  `input`/`output` are host or device pointers exactly as CompressBlocks* accepts
 * them (device pointers must be reachable from every GPU that takes a band, i.e. single-GPU or managed memory).
 * Blocks until the whole surface is encoded; returns true like the reference. */
bool CompressImageMT(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format);
bool CompressImageST(const rgba_surface* input, uint8_t* output, CompressionFunc* cmpFunc, int dxgi_format);

/* win32Threads.h:58-80 / win32Threads.cpp:289-329: profile trampolines */
void CompressImageBC1(const rgba_surface* input, uint8_t* output);
void CompressImageBC3(const rgba_surface* input, uint8_t* output);
/* same shape for the two formats the plugin sends to DirectX::Compress instead (IntelPlugin.cpp:271-273); with these the
 * band / slice rules above also keep the partial last block row and column (itw_bc45.h) */
void CompressImageBC4(const rgba_surface* input, uint8_t* output);
void CompressImageBC5(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_ultrafast(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_veryfast(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_fast(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_basic(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_slow(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_alpha_ultrafast(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_alpha_veryfast(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_alpha_fast(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_alpha_basic(const rgba_surface* input, uint8_t* output);
void CompressImageBC7_alpha_slow(const rgba_surface* input, uint8_t* output);
void CompressImageBC6H_veryfast(const rgba_surface* input, uint8_t* output);
void CompressImageBC6H_fast(const rgba_surface* input, uint8_t* output);
void CompressImageBC6H_basic(const rgba_surface* input, uint8_t* output);
void CompressImageBC6H_slow(const rgba_surface* input, uint8_t* output);
void CompressImageBC6H_veryslow(const rgba_surface* input, uint8_t* output);

/* The plugin's slice loop with progress / early out (IntelPlugin.cpp:851-879): the surface is cut into
 * `slices = width*height / slice_pixels` (>= 1) runs of block rows (rows [i*h/slices & ~3, (i+1)*h/slices & ~3), the reference's
 * arithmetic), `progress(i, slices, user)` is called for i = 1 .. slices-1 in order and aborts the job when it returns false (the
 * call then returns false).  `target` is the block array with `block_row_pitch` bytes between block rows (the reference passes the
 * DDS image's rowPitch); slice_pixels <= 0 selects the reference's 0x40000.
 *
 * The reference encodes slice i between progress(i) and progress(i+1), one synchronous CompressImageMT/ST call each.  Here, when
 * `cmpFunc` is one of THIS library's CompressImage* trampolines, the slices run as a PIPELINE instead:
 * W consecutive slices form a window (upload, kernels and download of neighbouring windows overlap on three streams; W =
 * itwSliceWindow(...): about 131 072 blocks, 262 144 for BC1/BC3/BC4/BC5 -- 0.3-1 ms of work, so any job long enough to show a progress bar
 * has many windows), and progress(i) is called once slice i-1 -- and
 * every slice before it -- is in `target`.  What a caller can observe of the difference:
 *   * when progress(i) returns false, slices < i are written like in the reference, and so may be up to W-1 slices after them (the
 *     rest of slice i-1's window); the window being encoded at that moment is drained and NOT copied back;
 *   * progress calls of one window arrive back to back;
 *   * `progress` runs on the calling thread while later windows are in flight on that thread's streams and staging buffers: it must not
 *     call back into this library on the same thread (Photoshop's SetProgress does not).
 * Several GPUs (`multithreaded` and GetProcessorCount() > 1, host memory): one pipeline PER GPU -- window k runs on worker k % n, each worker
 * on its own device, and the calling thread calls progress(i) once every window up to slice i-1's has arrived, i.e. still in order; on an
 * abort windows other GPUs had already finished further down the image stay written too.
 * Any other `cmpFunc` (a caller's own function: opaque), itwSetSliceWindow(-1) or a single slice: the literal loop.  Host or device
 * pointers; synchronous either way. */
typedef bool (ItwProgressFunc)(int done, int total, void* user);
bool itwCompressImageSliced(const rgba_surface* source, uint8_t* target, int64_t block_row_pitch, CompressionFunc* cmpFunc,
                            int dxgi_format, bool multithreaded, int64_t slice_pixels, ItwProgressFunc* progress, void* user);
/* The pipeline itself, for callers that hold a settings struct rather than a trampoline: `settings` = bc7_enc_settings* (BC7),
 * bc6h_enc_settings* (BC6H), ignored otherwise.  Same slices, progress contract and return value. */
bool itwCompressImageSlicedEx(const rgba_surface* source, uint8_t* target, int64_t block_row_pitch, int dxgi_format, const void* settings,
                              int64_t slice_pixels, ItwProgressFunc* progress, void* user);
/* W: slices per window.  itwSetSliceWindow(n > 0) fixes it process-wide (1 = the reference's early-out granularity exactly), 0 = by
 * format and size (default; env ITW_SLICE_WINDOW presets it), n < 0 = no pipeline: itwCompressImageSliced runs the literal loop
 * (also env ITW_SLICED_PIPELINE=0).  itwSliceWindow reports what a call would use (0: the literal loop). */
void itwSetSliceWindow(int slices);
int  itwSliceWindow(int dxgi_format, int width, int height, int64_t slice_pixels);
/* ... for given settings (bc7_enc_settings* / bc6h_enc_settings* / NULL): BC7 settings whose modes 1/3 scan every two-subset shape (`slow`, `alpha_slow`:
 * twice the work per block) take windows twice as large; itwSliceWindow is this with NULL (every preset the plugin selects). */
int  itwSliceWindowFor(int dxgi_format, const void* settings, int width, int height, int64_t slice_pixels);

/* Pad to multiples of 4 by edge replication (IntelPlugin.cpp:893-928): the step immediately before the ABI.
 * pixel_size = 4 (RGBA8) or 8 (RGBA16F).  Host version: returns a surface whose ptr was allocated with malloc()
 * (free with itwFreeSurface); the reference allocates with new[] and leaves ownership to the caller likewise.
 * Device version: `out_ptr` is a caller-allocated device buffer of ((w+3)&~3)*pixel_size x ((h+3)&~3) bytes, tight
 * pitch; asynchronous on the calling thread's stream (itwSetStream). */
rgba_surface itwPadToMultipleOf4(const rgba_surface* input, int pixel_size);
void itwFreeSurface(rgba_surface* s);
void itwPadToMultipleOf4Device(const rgba_surface* input, int pixel_size, uint8_t* out_ptr);

/* Photoshop's interleaved planes -> the encoder's surface, on the device (IntelPlugin.cpp:741-810 ConvertToBCFrom8/16/
 * 32Bit and :291-366 ConvertToBC6From8/16/32Bit): `src` holds width*height pixels of `planes` (1..4) interleaved
 * channels of `depth` bits (8, 16 = Photoshop's 0..32768 range, 32 = float); missing colour planes become 0, alpha is
 * opaque unless has_alpha (then plane 3; the reference's 32-bit -> half variant reads plane 2, kept).  `dst` is a
 * tightly pitched RGBA8 / RGBA16F surface.  Device pointers, asynchronous on the calling thread's stream.
 * gamma_correct applies the reference's pow(v, 1/2.2) to 32-bit input.  Returns 0, -1 on bad arguments. */
int itwConvertToRGBA8Device(const void* src, int depth, int planes, int has_alpha, int gamma_correct, int width, int height, uint8_t* dst);
int itwConvertToRGBA16FDevice(const void* src, int depth, int planes, int has_alpha, int width, int height, uint16_t* dst);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
