// host_rt.hpp -- host-side runtime helpers shared by the translation units of libispc_texcomp.so (defined in abi.hip).
//
// Error model.  The reference ABI is `void` and cannot fail (ispc_texcomp.h:104-107), and there is deliberately no CPU
// implementation behind this library.  A HIP failure (no device, out of memory, bad pointer) therefore has two
// possible outcomes, chosen by the host (itwSetErrorMode / ITW_ON_ERROR):
//   abort  (default)  diagnostic on stderr, abort() -- loud, never a silently wrong texture;
//   return            the ABI call returns without touching further state, the message is kept per host thread
//                     (itwLastError) and the bool-returning dispatch entry points return false.  A plug-in host that
//                     must not lose the user's document probes itwAvailable() first and checks itwLastError() after.
// Inside the library a failure is a C++ exception (itw::Failure) that every extern "C" entry point catches.
#pragma once
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/ispc_texcomp.h"

namespace itw {

struct Failure { char msg[384]; };

[[noreturn]] void fail_hip(const char* what, hipError_t e, const char* file, int line);
[[noreturn]] void fail_msg(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void report_failure(const Failure& f) noexcept;      // records the message; aborts in abort mode
void clear_failure() noexcept;

#define ITW_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) ::itw::fail_hip(#expr, e_, __FILE__, __LINE__); } while (0)

// Runs `body`; a Failure ends in report_failure().  Returns true on success.
template <class F>
inline bool guarded(F&& body) noexcept
{
    try { body(); return true; }
    catch (const Failure& f) { report_failure(f); }
    catch (...) { Failure f; std::snprintf(f.msg, sizeof f.msg, "unexpected C++ exception inside libispc_texcomp"); report_failure(f); }
    return false;
}

// One worker's share of a pipelined slice loop (abi.hip compress_sliced; dispatch.hip deals the windows of itwCompressImageSliced to the pool's
// GPUs): the worker runs windows part, part + parts, ... on ITS device and streams, calls `retired(first_slice, end_slice, ctx)` when a window's
// bytes are in the target instead of polling the caller's progress function (the submitting thread does that, in order), and stops issuing
// once `*stop` is set.
struct SlicedPart {
    int part = 0, parts = 1;
    void (*retired)(int first_slice, int end_slice, void* ctx) = nullptr;
    const std::atomic<bool>* stop = nullptr;
    void* ctx = nullptr;
};
// the share `part` of the slice loop over `source` (format, settings and slice size as itwCompressImageSlicedEx takes them); false: stopped or failed
bool sliced_part(const rgba_surface* source, uint8_t* target, int dxgi_format, const void* settings, int64_t slice_pixels, const SlicedPart& part);
// windows a call is cut into, and W (slices per window); 0 windows: the pipeline is off
int sliced_windows(int dxgi_format, const void* settings, int width, int height, int64_t slice_pixels, int* window_slices);

// One predicate for "the kernels can dereference this pointer": device AND managed allocations (abi.hip and
// dispatch.hip/decode.hip used to disagree on managed memory).  Unregistered host memory -> false.
bool is_device_pointer(const void* p) noexcept;

} // namespace itw
