// bc4_bc5.hip -- BC4_UNORM / BC5_UNORM encoders for gfx950.
//
// These two formats are the only ones the plugin does not send through kernel.ispc: IntelPlugin.cpp:271-273 hands the
// RGBA8 scratch image to DirectXTex (DirectX::Compress, TEX_COMPRESS_DEFAULT).  This file is the device replacement of
// that call: same block walk and partial-block rule (DirectXTexCompress.cpp:105-183), same endpoint optimiser
// (BC.h:727-856 OptimizeAlpha<false>, BC4BC5.cpp:186-238 FindEndPointsBC4U) and index choice (BC4BC5.cpp:314-337),
// evaluated in IEEE fp32 exactly as written, one rounding per operation (the library is built with -ffp-contract=off).
//
// Mapping: one lane per 8-byte channel block (BC5: lanes 2b and 2b+1 encode R and G of block b), so a wavefront writes
// 512 contiguous bytes; each lane keeps its 16 texel codes in 8 registers (two per dword) and runs the (at most eight)
// Newton iterations with a per-lane exit.
//
// Round 4 -- issue cycles, not instructions:
//   * FindClosestUNORM as RUN LENGTHS.  The reference searches 8 decoded levels per texel for the first strictly smallest
//     |level - texel| (16 x 8 x subtract / compare / two selects).  With 8-bit texels the function (red_0, red_1, v) -> index
//     has a domain of 256^3 cases, and for every endpoint pair it is piecewise constant in v with AT MOST 8 runs
//     (tests/test_bc45_index_table.py walks all 16.7 M cases against the reference's own FindClosestUNORM compiled from
//     BC4BC5.cpp; the pairs with red_0 == red_1, where the interpolated levels differ from the endpoints by an ulp, have at
//     most 5).  A kernel (bc45_build_index_table, once per device) runs the search AS WRITTEN for all 65 536 pairs and
//     stores, per pair, the 7 run starts and the 8 run indices (16 B; 1 MiB per device, L2 resident).  A block then loads ONE
//     entry and a texel's index is "how many run starts are <= v", looked up in the run-index word: two texels per dword,
//     add / and / add per run start on 2-cycle VOP2 forms, no float, no compare, no select.
//   * The Newton loop keeps a step's weights {c, d, c*c, d*d} (BC.h:729-732 quotients and their products: the same fp32
//     multiplications, done once) in one 16-byte LDS entry addressed by the step itself: the scale carries a factor 16
//     (exact: a power of two commutes with every rounding involved), so med3 / add / convert / and yield the entry's byte
//     offset.  Waves without a 6-step block (no texel 0 or 255 in 64 blocks) run a loop without the 6-step cases.
//   * Minimum / maximum of the 16 codes on packed halves (v_pk_min/max_u16: two texels per instruction), also for the 6-step
//     ramp's "smallest code above 0 / largest below 255" ((v + 255) & 255 and (v + 1) & 255 move the excluded code to the
//     other end of the range).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <mutex>
#include <vector>
#include "kernels.hpp"
#include "host_rt.hpp"

namespace itw {
namespace {

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_u16x2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_min_u16(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_min(as_u16x2(a), as_u16x2(b))); }
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) { return as_u32(__builtin_elementwise_max(as_u16x2(a), as_u16x2(b))); }

constexpr float UNORM_SCALE = 1.0f / 255.0f;                     // XMLoadUByteN4's SSE path: integer * (1/255)

// ---- FindClosestUNORM, tabulated ---------------------------------------------------------------------------------------------
// BC4_UNORM::DecodeFromIndex (BC4BC5.cpp:50-72), as written
__device__ __forceinline__ void decoded_levels(float (&g)[8], uint32_t r0, uint32_t r1)
{
    const float f0 = (float)r0 / 255.0f, f1 = (float)r1 / 255.0f;
    g[0] = f0; g[1] = f1;
    const bool eight = r0 > r1;
#pragma unroll
    for (int k = 1; k <= 6; k++) {
        const float v8 = (f0 * (float)(7 - k) + f1 * (float)k) / 7.0f;
        const float v6 = (k <= 4) ? (f0 * (float)(5 - k) + f1 * (float)k) / 5.0f : (k == 5 ? 0.0f : 1.0f);
        g[k + 1] = eight ? v8 : v6;
    }
}

// Entry of the run table for one endpoint pair: the index is constant on runs [T_j, T_j+1) of the texel code v, T_0 = 0.
//   x, y   C_j = 256 - T_j for run j = 1..7, one byte each (C_1 in the low byte of x; 0 = no such run): v + C_j carries into
//          bit 8 exactly when v >= T_j.  The top byte of y holds the number of runs (> 8 = the representation does not hold:
//          checked once after the build)
//   z, w   index of run j in byte j (z: runs 0..3, w: runs 4..7): a v_perm_b32 with run counts as selector bytes looks up four
//          texels at once
__global__ void __launch_bounds__(256) bc45_build_index_table(uint4* __restrict__ table)
{
    const uint32_t r0 = blockIdx.x, r1 = threadIdx.x;
    float g[8];
    decoded_levels(g, r0, r1);
    uint32_t c[2] = {0u, 0u}, order[2] = {0u, 0u}, runs = 0u, prev = 8u;
    for (uint32_t v = 0; v < 256u; v++) {
        const float t = (float)v * UNORM_SCALE;
        // FindClosestUNORM (BC4BC5.cpp:314-337): first index with the strictly smallest |g - t|
        uint32_t best = 0;
        float best_delta = 100000.0f;
#pragma unroll
        for (uint32_t k = 0; k < 8u; k++) {
            const float d = fabsf(g[k] - t);
            if (d < best_delta) { best = k; best_delta = d; }
        }
        if (best != prev) {
            if (runs >= 1u && runs < 8u) c[(runs - 1u) >> 2] |= (256u - v) << (8u * ((runs - 1u) & 3u));
            if (runs < 8u) order[runs >> 2] |= best << (8u * (runs & 3u));
            runs++;
            prev = best;
        }
    }
    table[r0 * 256u + r1] = make_uint4(c[0], c[1] | (min(runs, 255u) << 24), order[0], order[1]);
}

struct TableSlot { std::once_flag once; uint4* table = nullptr; };
TableSlot g_tables[64];

// Built once per device on a stream of the library's own (not on the caller's -- the first call used to synchronise the user's
// stream): the first BC4 / BC5 call on a device, or itwWarmupBC45(), blocks its HOST thread until the table exists; the caller's stream is
// never synchronised and later calls are plain asynchronous launches.  That one call allocates device memory, so it cannot run inside a
// stream capture: warm up first (include/itw_bc45.h).
const uint4* index_table()
{
    int dev = 0;
    ITW_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) fail_msg("BC4/BC5: device ordinal %d out of range", dev);
    TableSlot& s = g_tables[dev];
    std::call_once(s.once, [&] {                                  // an exception leaves the flag unset: the next call retries
        uint4* t = nullptr;
        hipStream_t own = nullptr;
        ITW_CHECK(hipStreamCreateWithFlags(&own, hipStreamNonBlocking));
        hipError_t e = hipMalloc(&t, 65536 * sizeof(uint4));
        if (e != hipSuccess) { (void)hipStreamDestroy(own); fail_hip("hipMalloc (BC4/BC5 index table)", e, __FILE__, __LINE__); }
        hipLaunchKernelGGL(bc45_build_index_table, dim3(256), dim3(256), 0, own, t);
        std::vector<uint4> h(65536);
        e = hipMemcpyAsync(h.data(), t, 65536 * sizeof(uint4), hipMemcpyDeviceToHost, own);
        if (e == hipSuccess) e = hipStreamSynchronize(own);       // complete before any later launch, on whatever stream
        (void)hipStreamDestroy(own);
        if (e != hipSuccess) { (void)hipFree(t); fail_hip("bc45_build_index_table", e, __FILE__, __LINE__); }
        for (const uint4& v : h)
            if ((v.y >> 24) < 1u || (v.y >> 24) > 8u) { (void)hipFree(t); fail_msg("BC4/BC5 index table: an endpoint pair has %u runs (at most 8 expected)", v.y >> 24); }
        s.table = t;
    });
    return s.table;
}

// ---- OptimizeAlpha<false> (BC.h:727-856) ---------------------------------------------------------------------------------------
// Step weights in LDS: entry s of table STEPS holds {c, d, c*c, d*d} with c = pC[s], d = pD[s] (BC.h:729-732: (STEPS-1-s)/(STEPS-1)
// and s/(STEPS-1) as compile-time fp32 quotients).  Entries 6 and 7 of the 6-step table are zero: the reference skips the codes
// "exactly 0" / "exactly 1" in its sums (`if (iStep < cSteps)`), and adding +-0 to an accumulator that started at +0 and only
// ever received sums of finite terms changes nothing (x + (+-0) = x; (+0) + (-0) = +0).
struct StepWeights { float c, d, cc, dd; };

__device__ __forceinline__ StepWeights step_weights(int steps, int s)
{
    // the quotients as the reference's tables hold them: k/7 and k/5 (k * (1/7) is not bit-equal to k/7 for k = 3 and 6)
    const float w8[8] = {0.0f / 7.0f, 1.0f / 7.0f, 2.0f / 7.0f, 3.0f / 7.0f, 4.0f / 7.0f, 5.0f / 7.0f, 6.0f / 7.0f, 7.0f / 7.0f};
    const float w6[6] = {0.0f / 5.0f, 1.0f / 5.0f, 2.0f / 5.0f, 3.0f / 5.0f, 4.0f / 5.0f, 5.0f / 5.0f};
    StepWeights r{0.f, 0.f, 0.f, 0.f};
    if (steps == 8)  { r.c = w8[7 - s]; r.d = w8[s]; }
    else if (s < 6)  { r.c = w6[5 - s]; r.d = w6[s]; }
    r.cc = r.c * r.c; r.dd = r.d * r.d;
    return r;
}

// One pass over the ramp for the lanes that are still iterating.  SIX = false: no lane of the wave has a 6-step block.
// `six` (per lane) selects the 6-step ramp; t[] are the texels as floats.
template <bool SIX>
__device__ __forceinline__ void newton_loop(float& fx, float& fy, const float (&t)[16], bool six, const StepWeights* tab8, const StepWeights* tab6)
{
    const float fsteps = (SIX && six) ? 5.0f : 7.0f;
    const float top16 = fsteps * 16.0f;
    const char* base = reinterpret_cast<const char*>((SIX && six) ? tab6 : tab8);
#pragma unroll 1
    for (int it = 0; it < 8; it++) {
        if ((fy - fx) < (1.0f / 256.0f)) break;
        const float scale = fsteps / (fy - fx);
        const float scale16 = scale * 16.0f;                      // exact: the step below comes out multiplied by 16 = its table offset
        // 6-step ramp: below the ramp the texel goes to "exactly 0" when it is nearer, above it to "exactly 1" (BC.h:789-804);
        // for 8-step lanes the two bounds are out of the texels' range
        const float to_zero = (SIX && six) ? fx * 0.5f : -1.0f;
        const float to_one  = (SIX && six) ? (fy + 1.0f) * 0.5f : 2.0f;
        float dx = 0.0f, dy = 0.0f, d2x = 0.0f, d2y = 0.0f;
        // four texels per batch: their steps first, then their four table reads in flight together, then the sums in texel order
        // (left to the compiler, every read was followed by its wait: sixteen serial LDS round trips per pass)
#pragma unroll
        for (int i0 = 0; i0 < 16; i0 += 4) {
            uint32_t off[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int i = i0 + u;
                const float dot16 = (t[i] - fx) * scale16;
                // dot <= 0 -> step 0, dot >= fsteps -> the last step, else (int)(dot + 0.5): clamp first, the clamped ends convert to themselves
                const float m = __builtin_amdgcn_fmed3f(dot16, 0.0f, top16);
                off[u] = (uint32_t)(int32_t)(m + 8.0f) & 0x70u;
                if (SIX) {
                    if (dot16 <= 0.0f && t[i] <= to_zero) off[u] = 6u * 16u;
                    if (dot16 >= top16 && t[i] >= to_one) off[u] = 7u * 16u;
                }
            }
            StepWeights w[4];
#pragma unroll
            for (int u = 0; u < 4; u++) w[u] = *reinterpret_cast<const StepWeights*>(base + off[u]);
            __builtin_amdgcn_sched_barrier(0);                    // keep the four reads ahead of the arithmetic that consumes them
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const float diff = (w[u].c * fx + w[u].d * fy) - t[i0 + u];
                dx += w[u].c * diff;
                d2x += w[u].cc;
                dy += w[u].d * diff;
                d2y += w[u].dd;
            }
        }
        if (d2x > 0.0f) fx -= dx / d2x;
        if (d2y > 0.0f) fy -= dy / d2y;
        if (fx > fy) { const float f = fx; fx = fy; fy = f; }
        if ((dx * dx < (1.0f / 64.0f)) && (dy * dy < (1.0f / 64.0f))) break;
    }
}

// One channel of one block -> 8 bytes (D3DXEncodeBC4U, BC4BC5.cpp:403-421).  P[j] = code of texel 2j | code of texel 2j+1 << 16.
__device__ __forceinline__ uint2 encode_channel(const uint32_t (&P)[8], const StepWeights* tab8, const StepWeights* tab6, const uint4* __restrict__ runs)
{
    float t[16];
#pragma unroll
    for (int j = 0; j < 8; j++) {                                 // one v_cvt_f32_ubyteN per texel, straight from the packed pair
        float lo, hi;
        asm("v_cvt_f32_ubyte0 %0, %1" : "=v"(lo) : "v"(P[j]));
        asm("v_cvt_f32_ubyte2 %0, %1" : "=v"(hi) : "v"(P[j]));
        t[2 * j] = lo * UNORM_SCALE;
        t[2 * j + 1] = hi * UNORM_SCALE;
    }
    // smallest / largest code, and for the 6-step ramp the smallest above 0 / largest below 255 (BC.h:742-766; fX starts at
    // 1.0 = code 255 and fY at 0.0 = code 0, which is also what "no such texel" leaves behind)
    uint32_t mn = P[0], mx = P[0];
#pragma unroll
    for (int j = 1; j < 8; j++) { mn = pk_min_u16(mn, P[j]); mx = pk_max_u16(mx, P[j]); }
    const uint32_t cmin = min(mn & 0xffffu, mn >> 16), cmax = max(mx & 0xffffu, mx >> 16);
    const bool six = (cmin == 0u) || (cmax == 255u);                     // BC4BC5.cpp:213: 0.0f == fBlockMin || 1.0f == fBlockMax
    const bool any_six = __any(six);                                     // wave-uniform: most waves have no 6-step block at all
    uint32_t cx = cmin, cy = cmax;
    if (any_six) {
        uint32_t lo6 = (P[0] + 0x00ff00ffu) & 0x00ff00ffu, hi6 = (P[0] + 0x00010001u) & 0x00ff00ffu;
#pragma unroll
        for (int j = 1; j < 8; j++) {
            lo6 = pk_min_u16(lo6, (P[j] + 0x00ff00ffu) & 0x00ff00ffu);   // (v - 1) mod 256: code 0 becomes the largest
            hi6 = pk_max_u16(hi6, (P[j] + 0x00010001u) & 0x00ff00ffu);   // (v + 1) mod 256: code 255 becomes the smallest
        }
        if (six) {
            const uint32_t l = min(lo6 & 0xffffu, lo6 >> 16), h = max(hi6 & 0xffffu, hi6 >> 16);
            cx = min(l + 1u, 255u);
            cy = h ? h - 1u : 0u;
        }
    }
    float fx = (float)cx * UNORM_SCALE, fy = (float)cy * UNORM_SCALE;
    if (six && fx == fy) fy = 1.0f;
    if (any_six) newton_loop<true>(fx, fy, t, six, tab8, tab6);
    else            newton_loop<false>(fx, fy, t, false, tab8, tab6);
    const float ox = (fx < 0.0f) ? 0.0f : (fx > 1.0f) ? 1.0f : fx;
    const float oy = (fy < 0.0f) ? 0.0f : (fy > 1.0f) ? 1.0f : fy;
    // BC4BC5.cpp:213-237: the 8-step ramp stores (max, min), the 6-step ramp (min, max)
    const uint32_t qx = (uint32_t)(ox * 255.0f), qy = (uint32_t)(oy * 255.0f);
    const uint32_t r0 = six ? qx : qy, r1 = six ? qy : qx;

    // FindClosestUNORM through the run table: count the run starts that are <= v, two texels per dword
    const uint4 e = runs[(r0 << 8) | r1];
    uint32_t C[7];
#pragma unroll
    for (int j = 0; j < 7; j++)
        C[j] = __builtin_amdgcn_perm(0u, j < 4 ? e.x : e.y, 0x0c000c00u | (uint32_t)((j & 3) * 0x00010001));   // byte j -> both halves
    // run indices as bytes (e.z: runs 0..3, e.w: runs 4..7): a v_perm_b32 whose selector bytes are the run counts (0..7) of FOUR
    // texels looks all four up at once
    const uint32_t tab_lo = e.z, tab_hi = e.w;
    const uint32_t carry = 0x01000100u;
    uint32_t part[2] = {0u, 0u};                                  // 3-bit indices of texels 0..7 / 8..15
#pragma unroll
    for (int q = 3; q >= 0; q--) {                                // four texels (two pairs) per step, from the last down: (acc << 12) | 12 bits
        uint32_t sa = (P[2 * q] + C[0]) & carry, sb = (P[2 * q + 1] + C[0]) & carry;
#pragma unroll
        for (int k = 1; k < 7; k++) { sa += (P[2 * q] + C[k]) & carry; sb += (P[2 * q + 1] + C[k]) & carry; }
        // the counts sit in bytes 1 and 3 of each sum: gather them in texel order, look the run indices up, squeeze 4 x 3 bits together
        const uint32_t counts = __builtin_amdgcn_perm(sb, sa, 0x07050301u);
        const uint32_t idx = __builtin_amdgcn_perm(tab_hi, tab_lo, counts);
        const uint32_t two = (idx | (idx >> 5)) & 0x003f003fu;    // texels (0,1) in bits 0..5, (2,3) in bits 16..21
        const uint32_t four = (two | (two >> 10)) & 0x0fffu;
        uint32_t& acc = part[q >> 1];
        acc = (acc << 12) | four;
    }
    return make_uint2(r0 | (r1 << 8) | (part[0] << 16), (part[0] >> 16) | (part[1] << 8));
}

// register budget (waves per SIMD) and grid, tuning constants like the BC7 kernels'
#ifndef BC45_WAVES
#define BC45_WAVES 4
#endif
#ifndef BC45_GRID
#define BC45_GRID 2048
#endif

// The sixteen texel words of channel-block `lane` (4 x dwordx4 when VEC16 and the block is whole).  Partial blocks: missing columns /
// rows repeat source column / row {0,0,0,1}[i], itself wrapped to 0 when that one is missing too (DirectXTexCompress.cpp:140-168
// applied in its own order).
template <int NCH, bool VEC16>
__device__ __forceinline__ void load_words45(uint32_t (&w)[16], const uint8_t* __restrict__ src, int64_t stride, int32_t width, int32_t height,
                                             int32_t blocks_x, int32_t lane)
{
    const int32_t b = (NCH == 2) ? (lane >> 1) : lane;
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    const int32_t pw = min(4, width - 4 * xx), ph = min(4, height - 4 * yy);
    const uint8_t* p = src + (int64_t)yy * 4 * stride + (int64_t)xx * 16;
    if (pw == 4 && ph == 4) {
#pragma unroll
        for (int y = 0; y < 4; y++) {
            if (VEC16) {
                const uint4 v = *reinterpret_cast<const uint4*>(p + y * stride);
                w[4 * y] = v.x; w[4 * y + 1] = v.y; w[4 * y + 2] = v.z; w[4 * y + 3] = v.w;
            } else {
                const uint32_t* q = reinterpret_cast<const uint32_t*>(p + y * stride);
                w[4 * y] = q[0]; w[4 * y + 1] = q[1]; w[4 * y + 2] = q[2]; w[4 * y + 3] = q[3];
            }
        }
    } else {
#pragma unroll
        for (int y = 0; y < 4; y++) {
            int sy = y < ph ? y : (y == 3 ? 1 : 0);
            if (sy >= ph) sy = 0;
#pragma unroll
            for (int x = 0; x < 4; x++) {
                int sx = x < pw ? x : (x == 3 ? 1 : 0);
                if (sx >= pw) sx = 0;
                w[4 * y + x] = *reinterpret_cast<const uint32_t*>(p + sy * stride + sx * 4);
            }
        }
    }
}

// Persistent workgroups (round 4): a wave of this kernel executes ~630 VALU instructions per block -- a fifth of it waiting for its
// own texels if it loads, encodes, stores and exits (measured: 55-62 % of the issue slots used).  So at most BC45_GRID workgroups walk
// the surface in chunks of 256 channel blocks and request the NEXT chunk's texels before encoding the current one.
// WHOLE (the common case: width and height multiples of 4, the surface below 2 GiB): no partial blocks, and the walk keeps each
// lane's block position and byte offset incrementally -- one division per workgroup instead of one per chunk, 32-bit offsets from the
// uniform base instead of four 64-bit row pointers with their source-row / column selects (about 90 of the 630 instructions).
template <int NCH, bool VEC16, bool WHOLE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BC45_WAVES, BC45_WAVES)))
bc45_kernel(const uint8_t* __restrict__ src, int64_t stride, int32_t width, int32_t height, int32_t blocks_x,
            int32_t nlanes, uint8_t* __restrict__ dst, const uint4* __restrict__ runs)
{
    __shared__ StepWeights s_tab[16];                             // [0..7] the 8-step ramp, [8..15] the 6-step ramp
    int32_t lane = blockIdx.x * 256 + threadIdx.x;
    uint32_t w[16];
    // WHOLE: block position of this lane and its byte offset from `src`; per chunk both advance by workgroup-uniform steps
    const int32_t step_blocks = (int32_t)gridDim.x * (NCH == 2 ? 128 : 256);
    const int32_t step_y = step_blocks / blocks_x, step_x = step_blocks - step_y * blocks_x;       // scalar, once
    const uint32_t stride32 = (uint32_t)stride;
    int32_t xx = 0;
    uint32_t off = 0;
    auto load_whole = [&](uint32_t o) {
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const uint8_t* p = src + (o + (uint32_t)y * stride32);
            if (VEC16) {
                const uint4 v = *reinterpret_cast<const uint4*>(p);
                w[4 * y] = v.x; w[4 * y + 1] = v.y; w[4 * y + 2] = v.z; w[4 * y + 3] = v.w;
            } else {
                const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
                w[4 * y] = q[0]; w[4 * y + 1] = q[1]; w[4 * y + 2] = q[2]; w[4 * y + 3] = q[3];
            }
        }
    };
    if (WHOLE) {
        const int32_t l0 = lane < nlanes ? lane : nlanes - 1;
        const int32_t b0 = (NCH == 2) ? (l0 >> 1) : l0;
        const int32_t yy = b0 / blocks_x;
        xx = b0 - yy * blocks_x;
        off = (uint32_t)yy * 4u * stride32 + (uint32_t)xx * 16u;
        load_whole(off);
    } else {
        load_words45<NCH, VEC16>(w, src, stride, width, height, blocks_x, lane < nlanes ? lane : nlanes - 1);
    }
    if (threadIdx.x < 16) s_tab[threadIdx.x] = step_weights(threadIdx.x < 8 ? 8 : 6, threadIdx.x & 7);
    __syncthreads();
    const uint32_t ch = (NCH == 2) ? (uint32_t)(threadIdx.x & 1) : 0u;   // byte of the RGBA8 word this lane encodes (the grid step is even)
    const uint32_t sel = 0x0c040c00u + ch * 0x00010001u;          // v_perm: byte ch of the first word | byte ch of the second << 16
    const int32_t step = (int32_t)gridDim.x * 256;
    const uint32_t off_step = (uint32_t)step_y * 4u * stride32 + (uint32_t)step_x * 16u;
    const uint32_t off_wrap = 4u * stride32 - (uint32_t)blocks_x * 16u;           // one block row down, blocks_x blocks back
    for (;;) {
        uint32_t P[8];
#pragma unroll
        for (int j = 0; j < 8; j++) P[j] = __builtin_amdgcn_perm(w[2 * j + 1], w[2 * j], sel);
        const int32_t cur = lane;
        lane += step;
        const bool more = (lane - (int32_t)threadIdx.x) < nlanes;         // uniform per workgroup: some lane of the next chunk exists
        if (more) {
            if (WHOLE) {
                xx += step_x; off += off_step;
                if (xx >= blocks_x) { xx -= blocks_x; off += off_wrap; }
                // lanes past the end (last chunk only) re-read the surface's last block: clamp the offset like the lane index
                const uint32_t last = (uint32_t)(height / 4 - 1) * 4u * stride32 + (uint32_t)(blocks_x - 1) * 16u;
                load_whole(lane < nlanes ? off : last);
            } else {
                load_words45<NCH, VEC16>(w, src, stride, width, height, blocks_x, lane < nlanes ? lane : nlanes - 1);
            }
        }
        const uint2 o = encode_channel(P, s_tab, s_tab + 8, runs);
        if (cur < nlanes) *reinterpret_cast<uint2*>(dst + (int64_t)cur * 8) = o;
        if (!more) break;
    }
}

template <int NCH>
void launch_bc45(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st)
{
    if (width <= 0 || height <= 0) return;
    const int bx = (width + 3) / 4, by = (height + 3) / 4;       // DirectXTex keeps partial blocks
    const int64_t n = (int64_t)bx * by * NCH;
    const bool vec = ((reinterpret_cast<uintptr_t>(src) | (uintptr_t)stride) & 15) == 0;
    const uint4* runs = index_table();
    const int64_t chunks = (n + 255) / 256;
    const dim3 grid((unsigned)(chunks < BC45_GRID ? chunks : BC45_GRID)), blk(256);
    // WHOLE: no partial blocks and every texel offset fits 31 bits (a non-negative stride; 16384^2 RGBA8 is 1 GiB)
    const bool whole = (width % 4 == 0) && (height % 4 == 0) && stride > 0 && (int64_t)height * stride < ((int64_t)1 << 31);
    if (whole) {
        if (vec) hipLaunchKernelGGL((bc45_kernel<NCH, true, true>),  grid, blk, 0, st, src, stride, width, height, bx, (int32_t)n, dst, runs);
        else     hipLaunchKernelGGL((bc45_kernel<NCH, false, true>), grid, blk, 0, st, src, stride, width, height, bx, (int32_t)n, dst, runs);
    } else {
        if (vec) hipLaunchKernelGGL((bc45_kernel<NCH, true, false>),  grid, blk, 0, st, src, stride, width, height, bx, (int32_t)n, dst, runs);
        else     hipLaunchKernelGGL((bc45_kernel<NCH, false, false>), grid, blk, 0, st, src, stride, width, height, bx, (int32_t)n, dst, runs);
    }
}

} // namespace

void launch_bc4(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st) { launch_bc45<1>(src, stride, width, height, dst, st); }
void launch_bc5(const uint8_t* src, int64_t stride, int width, int height, uint8_t* dst, hipStream_t st) { launch_bc45<2>(src, stride, width, height, dst, st); }

void warmup_bc45() { (void)index_table(); }

#ifdef ITW_TEST_HOOKS
// test hook (include/itw_test_hooks.h): the run table of the current device, 65 536 entries x 4 words, to host memory
void copy_bc45_index_table(uint32_t* host_out, hipStream_t st)
{
    const uint4* t = index_table();
    ITW_CHECK(hipMemcpyAsync(host_out, t, 65536 * sizeof(uint4), hipMemcpyDeviceToHost, st));
    ITW_CHECK(hipStreamSynchronize(st));
}
#endif

} // namespace itw
