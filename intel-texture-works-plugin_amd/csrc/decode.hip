// decode.hip -- BCn block decoders for gfx950 (include/itw_decode.h).  One block per lane: 8 / 16 B in, 64 B (RGBA8)
// or 128 B (RGBA16F) out as four row stores of 16 / 32 B per lane.  Pure integer work, HBM bound by the output
// (4-8x the input).  Written from the format definitions (the readable statement inside the reference tree is its
// decoder: BC.cpp for BC1/BC3, BC6HBC7.cpp:35-37 weights, :40 partitions, :247 fix-ups, :537 BC7 mode table,
// :1937-2140 BC7 decode, :310-500 BC6H mode descriptors, :1077-1210 BC6H decode, :1313-1359 unquantisation).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "../../include/itw_decode.h"
#include "../../include/itw_amd.h"
#include "bc6h_layout.hpp"
#include "host_rt.hpp"

namespace itw {

#define BCN_TABLE_QUAL __device__ const
namespace dec {
#include "bc7_tables.h"
}
#undef BCN_TABLE_QUAL

__device__ const Bc6hLayout D_BC6H_LAYOUT[14] = {
    BC6H_LAYOUT[0], BC6H_LAYOUT[1], BC6H_LAYOUT[2], BC6H_LAYOUT[3], BC6H_LAYOUT[4], BC6H_LAYOUT[5], BC6H_LAYOUT[6],
    BC6H_LAYOUT[7], BC6H_LAYOUT[8], BC6H_LAYOUT[9], BC6H_LAYOUT[10], BC6H_LAYOUT[11], BC6H_LAYOUT[12], BC6H_LAYOUT[13]};

__device__ const unsigned char D_WEIGHTS[3][16] = {
    {0, 21, 43, 64}, {0, 9, 18, 27, 37, 46, 55, 64}, {0, 4, 9, 13, 17, 21, 26, 30, 34, 38, 43, 47, 51, 55, 60, 64}};

// LSB-first reader over a 128-bit block
struct Bits {
    unsigned long long lo, hi;
    int pos;
    __device__ __forceinline__ uint32_t take(int n)
    {
        if (n == 0) return 0u;
        unsigned long long v;
        if (pos >= 64) v = hi >> (pos - 64);
        else v = (lo >> pos) | (pos ? (hi << (64 - pos)) : 0ull);
        pos += n;
        return (uint32_t)(v & ((1ull << n) - 1ull));
    }
};

// ---- BC1 colour block; punch-through (3-colour) mode only where the format allows it ----------------------
__device__ __forceinline__ void decode_color(uint32_t w0, uint32_t idx, bool allow3, uint32_t (&px)[16])
{
    const uint32_t c0 = w0 & 0xffffu, c1 = w0 >> 16;
    int pal[4][3];
    const uint32_t c[2] = {c0, c1};
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const int r = (c[i] >> 11) & 31, g = (c[i] >> 5) & 63, b = c[i] & 31;
        pal[i][0] = (r << 3) | (r >> 2); pal[i][1] = (g << 2) | (g >> 4); pal[i][2] = (b << 3) | (b >> 2);
    }
    const bool four = c0 > c1 || !allow3;
    uint32_t a2 = 255u, a3 = 255u;
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        pal[2][ch] = four ? (2 * pal[0][ch] + pal[1][ch] + 1) / 3 : (pal[0][ch] + pal[1][ch]) / 2;
        pal[3][ch] = four ? (pal[0][ch] + 2 * pal[1][ch] + 1) / 3 : 0;
    }
    if (!four) a3 = 0u;
    uint32_t packed[4];
#pragma unroll
    for (int i = 0; i < 4; i++)
        packed[i] = (uint32_t)pal[i][0] | ((uint32_t)pal[i][1] << 8) | ((uint32_t)pal[i][2] << 16) | ((i == 3 ? a3 : (i == 2 ? a2 : 255u)) << 24);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t q = (idx >> (2 * k)) & 3u;
        px[k] = q == 0 ? packed[0] : q == 1 ? packed[1] : q == 2 ? packed[2] : packed[3];
    }
}

// The 8-byte interpolated-scalar block shared by BC3 alpha, BC4 and both halves of BC5; the value lands in byte SHIFT/8.
template <int SHIFT>
__device__ __forceinline__ void decode_scalar_block(uint32_t w0, uint32_t w1, uint32_t (&px)[16])
{
    int a[8];
    a[0] = (int)(w0 & 255u); a[1] = (int)((w0 >> 8) & 255u);
    if (a[0] > a[1]) {
#pragma unroll
        for (int i = 1; i < 7; i++) a[1 + i] = ((7 - i) * a[0] + i * a[1] + 3) / 7;
    } else {
#pragma unroll
        for (int i = 1; i < 5; i++) a[1 + i] = ((5 - i) * a[0] + i * a[1] + 2) / 5;
        a[6] = 0; a[7] = 255;
    }
    const unsigned long long bits = ((unsigned long long)(w0 >> 16)) | ((unsigned long long)w1 << 16);
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint32_t q = (uint32_t)(bits >> (3 * k)) & 7u;
        int v = a[0];
#pragma unroll
        for (int i = 1; i < 8; i++) v = (q == (uint32_t)i) ? a[i] : v;
        px[k] = (px[k] & ~(0xffu << SHIFT)) | ((uint32_t)v << SHIFT);
    }
}
__device__ __forceinline__ void decode_bc3_alpha(uint32_t w0, uint32_t w1, uint32_t (&px)[16]) { decode_scalar_block<24>(w0, w1, px); }

// ---- BC7 ---------------------------------------------------------------------------------------------------
struct Bc7Mode { unsigned char ns, pb, rb, isb, cb, ab, epb, spb, ib, ib2; };
__device__ const Bc7Mode D_BC7_MODES[8] = {
    {3, 4, 0, 0, 4, 0, 1, 0, 3, 0}, {2, 6, 0, 0, 6, 0, 0, 1, 3, 0}, {3, 6, 0, 0, 5, 0, 0, 0, 2, 0}, {2, 6, 0, 0, 7, 0, 1, 0, 2, 0},
    {1, 0, 2, 1, 5, 6, 0, 0, 2, 3}, {1, 0, 2, 0, 7, 8, 0, 0, 2, 2}, {1, 0, 0, 0, 7, 7, 1, 0, 4, 0}, {2, 6, 0, 0, 5, 5, 1, 0, 2, 0}};

__device__ __forceinline__ int decode_bc7(Bits& bs, uint32_t (&px)[16])
{
    int mode = 0;
    while (mode < 8 && !bs.take(1)) mode++;
    if (mode == 8) {
#pragma unroll
        for (int k = 0; k < 16; k++) px[k] = 0u;
        return -1;
    }
    const Bc7Mode mi = D_BC7_MODES[mode];
    const int shape = (int)bs.take(mi.pb), rot = (int)bs.take(mi.rb), isel = (int)bs.take(mi.isb);
    int ep[6][4];
    for (int ch = 0; ch < 3; ch++)
        for (int e = 0; e < 6; e++) ep[e][ch] = (e < mi.ns * 2) ? (int)bs.take(mi.cb) : 0;
    for (int e = 0; e < 6; e++) ep[e][3] = (e < mi.ns * 2 && mi.ab) ? (int)bs.take(mi.ab) : 255;
    int cbits = mi.cb, abits = mi.ab;
    if (mi.epb) {
        for (int e = 0; e < 6; e++)
            if (e < mi.ns * 2) {
                const int p = (int)bs.take(1);
                for (int ch = 0; ch < 3; ch++) ep[e][ch] = (ep[e][ch] << 1) | p;
                if (mi.ab) ep[e][3] = (ep[e][3] << 1) | p;
            }
        cbits++; if (mi.ab) abits++;
    } else if (mi.spb) {
        for (int s = 0; s < 3; s++)
            if (s < mi.ns) {
                const int p = (int)bs.take(1);
                for (int e = 2 * s; e < 2 * s + 2; e++)
                    for (int ch = 0; ch < 3; ch++) ep[e][ch] = (ep[e][ch] << 1) | p;
            }
        cbits++;
    }
    for (int e = 0; e < 6; e++) {
        for (int ch = 0; ch < 3; ch++) { const int v = ep[e][ch] << (8 - cbits); ep[e][ch] = v | (v >> cbits); }
        if (mi.ab) { const int v = ep[e][3] << (8 - abits); ep[e][3] = v | (v >> abits); }
    }
    const int table = (mi.ns == 3) ? 64 + shape : shape;
    const uint32_t pattern = (mi.ns == 1) ? 0u : dec::BCN_PATTERN[table];
    const int anc1 = (mi.ns >= 2) ? (dec::BCN_ANCHORS[table] >> 4) : 0, anc2 = (mi.ns >= 2) ? (dec::BCN_ANCHORS[table] & 15) : 0;
    uint32_t i1[2] = {0u, 0u}, i2[2] = {0u, 0u};          // 4 bits per texel
    for (int k = 0; k < 16; k++) {
        const int sub = (int)((pattern >> (2 * k)) & 3u);
        const int anchor = sub == 0 ? 0 : (sub == 1 ? anc1 : anc2);
        const uint32_t v = bs.take(mi.ib - (k == anchor ? 1 : 0));
        i1[k >> 3] |= v << (4 * (k & 7));
    }
    if (mi.ib2)
        for (int k = 0; k < 16; k++) i2[k >> 3] |= bs.take(mi.ib2 - (k == 0 ? 1 : 0)) << (4 * (k & 7));
    for (int k = 0; k < 16; k++) {
        const int sub = (int)((pattern >> (2 * k)) & 3u);
        int ci = (int)((i1[k >> 3] >> (4 * (k & 7))) & 15u), ai = ci, cbw = mi.ib, abw = mi.ib;
        if (mi.ib2) {
            const int second = (int)((i2[k >> 3] >> (4 * (k & 7))) & 15u);
            if (isel) { ci = second; cbw = mi.ib2; } else { ai = second; abw = mi.ib2; }
        }
        const int wc = D_WEIGHTS[cbw - 2][ci], wa = D_WEIGHTS[abw - 2][ai];
        int v[4];
        for (int ch = 0; ch < 4; ch++) {
            int e0 = 0, e1 = 0;
            for (int s = 0; s < 3; s++) if (s == sub) { e0 = ep[2 * s][ch]; e1 = ep[2 * s + 1][ch]; }
            const int w = (ch == 3) ? wa : wc;
            v[ch] = (e0 * (64 - w) + e1 * w + 32) >> 6;
        }
        if (!mi.ab) v[3] = 255;
        if (rot == 1) { const int t = v[3]; v[3] = v[0]; v[0] = t; }
        else if (rot == 2) { const int t = v[3]; v[3] = v[1]; v[1] = t; }
        else if (rot == 3) { const int t = v[3]; v[3] = v[2]; v[2] = t; }
        px[k] = (uint32_t)v[0] | ((uint32_t)v[1] << 8) | ((uint32_t)v[2] << 16) | ((uint32_t)v[3] << 24);
    }
    return mode;
}

// ---- BC6H, unsigned ----------------------------------------------------------------------------------------
__device__ __forceinline__ int unquantize_uf16(int comp, int bits)
{
    if (bits >= 15) return comp;
    if (comp == 0) return 0;
    if (comp == ((1 << bits) - 1)) return 0xFFFF;
    return ((comp << 16) + 0x8000) >> bits;
}

__device__ __forceinline__ int decode_bc6h(Bits& bs, uint32_t (&lo)[16], uint32_t (&hi)[16])
{
    int m = (int)bs.take(2);
    if (m >= 2) m |= (int)bs.take(3) << 2;
    int mode = -1;
    for (int i = 0; i < 14; i++) if (D_BC6H_LAYOUT[i].prefix == m) mode = i;
    if (mode < 0) {
#pragma unroll
        for (int k = 0; k < 16; k++) { lo[k] = 0u; hi[k] = 0x3C000000u; }
        return -1;
    }
    const Bc6hLayout& L = D_BC6H_LAYOUT[mode];
    int e[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    int shape = 0;
    const int header = L.two_regions ? 82 : 65;
    while (bs.pos < header) {
        const int cur = bs.pos;
        if (bs.take(1)) {
            const int field = L.slot[cur] >> 4, bit = L.slot[cur] & 15;
            if (field == 2) shape |= 1 << bit;
            else if (field >= 3) {
                const int f = field - 3, which = f & 3, ch = f >> 2;
                for (int a = 0; a < 4; a++) for (int c = 0; c < 3; c++) if (a == which && c == ch) e[a][c] |= 1 << bit;
            }
        }
    }
    if (L.transformed)
        for (int ch = 0; ch < 3; ch++) {
            const int mask = (1 << L.base_bits[ch]) - 1, db = L.delta_bits[ch];
            for (int k = 1; k < 4; k++)
                if (k < (L.two_regions ? 4 : 2)) {
                    const int d = (e[k][ch] & (1 << (db - 1))) ? (e[k][ch] | ~((1 << db) - 1)) : e[k][ch];
                    e[k][ch] = (d + e[0][ch]) & mask;
                }
        }
    const int ib = L.two_regions ? 3 : 4;
    const uint32_t pattern = L.two_regions ? dec::BCN_PATTERN[shape] : 0u;
    const int anchor1 = L.two_regions ? (dec::BCN_ANCHORS[shape] >> 4) : -1;
    for (int k = 0; k < 16; k++) {
        const int region = (int)((pattern >> (2 * k)) & 3u);
        const int n = ib - ((k == 0 || (region == 1 && k == anchor1)) ? 1 : 0);
        const int w = D_WEIGHTS[ib - 2][bs.take(n)];
        int v[3];
        for (int ch = 0; ch < 3; ch++) {
            const int a = unquantize_uf16(region ? e[2][ch] : e[0][ch], L.base_bits[ch]);
            const int b = unquantize_uf16(region ? e[3][ch] : e[1][ch], L.base_bits[ch]);
            int x = (a * (64 - w) + b * w + 32) >> 6;
            x = (x * 31) >> 6;
            v[ch] = x > 0x7BFF ? 0x7BFF : x;
        }
        lo[k] = (uint32_t)v[0] | ((uint32_t)v[1] << 16);
        hi[k] = (uint32_t)v[2] | 0x3C000000u;
    }
    return mode;
}

// FMT: 1 BC1, 3 BC3, 7 BC7, 6 BC6H
template <int FMT>
__global__ void __launch_bounds__(256)
decode_kernel(const uint8_t* __restrict__ blocks, int32_t blocks_x, int32_t nblocks, uint8_t* __restrict__ out, int64_t stride,
              int32_t* __restrict__ modes, int32_t width, int32_t height)
{
    const int32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= nblocks) return;
    const int32_t yy = b / blocks_x, xx = b - yy * blocks_x;
    int mode = 0;
    if (FMT == 6) {
        const uint4 w = *reinterpret_cast<const uint4*>(blocks + (int64_t)b * 16);
        Bits bs{(unsigned long long)w.x | ((unsigned long long)w.y << 32), (unsigned long long)w.z | ((unsigned long long)w.w << 32), 0};
        uint32_t lo[16], hi[16];
        mode = decode_bc6h(bs, lo, hi);
        uint8_t* o = out + (int64_t)yy * 4 * stride + (int64_t)xx * 32;
#pragma unroll
        for (int y = 0; y < 4; y++) {
            uint32_t* row = reinterpret_cast<uint32_t*>(o + y * stride);
#pragma unroll
            for (int x = 0; x < 4; x++) { row[2 * x] = lo[y * 4 + x]; row[2 * x + 1] = hi[y * 4 + x]; }
        }
    } else {
        uint32_t px[16];
        if (FMT == 1) {
            const uint2 w = *reinterpret_cast<const uint2*>(blocks + (int64_t)b * 8);
            decode_color(w.x, w.y, true, px);
        } else if (FMT == 3) {
            const uint4 w = *reinterpret_cast<const uint4*>(blocks + (int64_t)b * 16);
            decode_color(w.z, w.w, false, px);
            decode_bc3_alpha(w.x, w.y, px);
        } else if (FMT == 4) {                                   // BC4_UNORM -> (R, 0, 0, 255) like D3DXDecodeBC4U (BC4BC5.cpp:373-385)
            const uint2 w = *reinterpret_cast<const uint2*>(blocks + (int64_t)b * 8);
#pragma unroll
            for (int k = 0; k < 16; k++) px[k] = 0xff000000u;
            decode_scalar_block<0>(w.x, w.y, px);
        } else if (FMT == 5) {                                   // BC5_UNORM -> (R, G, 0, 255) (BC4BC5.cpp:449-462)
            const uint4 w = *reinterpret_cast<const uint4*>(blocks + (int64_t)b * 16);
#pragma unroll
            for (int k = 0; k < 16; k++) px[k] = 0xff000000u;
            decode_scalar_block<0>(w.x, w.y, px);
            decode_scalar_block<8>(w.z, w.w, px);
        } else {
            const uint4 w = *reinterpret_cast<const uint4*>(blocks + (int64_t)b * 16);
            Bits bs{(unsigned long long)w.x | ((unsigned long long)w.y << 32), (unsigned long long)w.z | ((unsigned long long)w.w << 32), 0};
            mode = decode_bc7(bs, px);
        }
        uint8_t* o = out + (int64_t)yy * 4 * stride + (int64_t)xx * 16;
        // BC4 / BC5 streams may end in partial blocks (the encoder keeps them, itw_bc45.h): texels beyond the surface are cropped
        const int ny = (FMT == 4 || FMT == 5) ? min(4, height - yy * 4) : 4, nx = (FMT == 4 || FMT == 5) ? min(4, width - xx * 4) : 4;
#pragma unroll
        for (int y = 0; y < 4; y++) {
            if (y >= ny) break;
            uint32_t* row = reinterpret_cast<uint32_t*>(o + y * stride);
#pragma unroll
            for (int x = 0; x < 4; x++) if (x < nx) row[x] = px[y * 4 + x];
        }
    }
    if (modes) modes[b] = mode;
}

} // namespace itw

namespace {

bool on_device(const void* p) { return itw::is_device_pointer(p); }     // device and managed memory alike

#define DEC_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "itwDecodeBlocks: %s failed: %s; aborting\n", #x, hipGetErrorString(e_)); std::abort(); } } while (0)

} // namespace

extern "C" int itwDecodeBlocks(int f, const uint8_t* blocks, int width, int height, uint8_t* out, int64_t out_stride, int32_t* modes)
{
    const int kind = (f == 71 || f == 72) ? 1 : (f == 77 || f == 78) ? 3 : (f == 98 || f == 99) ? 7 : (f == 95 || f == 96) ? 6 : f == 80 ? 4 : f == 83 ? 5 : 0;
    const bool partial_ok = (kind == 4 || kind == 5);            // the DirectXTex formats keep partial blocks
    if (!kind || (out_stride & 3)) return -1;
    if (partial_ok ? (width < 1 || height < 1) : (width < 4 || height < 4 || (width & 3) || (height & 3))) return -1;
    const int bx = (width + 3) / 4, by = (height + 3) / 4;
    const int64_t n = (int64_t)bx * by;
    const size_t in_bytes = (size_t)n * ((kind == 1 || kind == 4) ? 8 : 16), texel = kind == 6 ? 8 : 4;
    const size_t row_bytes = (size_t)width * texel;
    if ((size_t)out_stride < row_bytes) return -1;
    hipStream_t st = (hipStream_t)itwGetStream();
    const bool din = on_device(blocks), dout = on_device(out), dmodes = !modes || on_device(modes);

    uint8_t *d_in = const_cast<uint8_t*>(blocks), *d_out = out;
    int32_t* d_modes = modes;
    int64_t d_stride = out_stride;
    if (!din)  { DEC_CHECK(hipMalloc((void**)&d_in, in_bytes)); DEC_CHECK(hipMemcpyAsync(d_in, blocks, in_bytes, hipMemcpyHostToDevice, st)); }
    if (!dout) { d_stride = (int64_t)row_bytes; DEC_CHECK(hipMalloc((void**)&d_out, row_bytes * (size_t)height)); }
    if (modes && !dmodes) DEC_CHECK(hipMalloc((void**)&d_modes, (size_t)n * 4));
    const dim3 grid((unsigned)((n + 255) / 256)), blk(256);
    switch (kind) {
    case 1: hipLaunchKernelGGL((itw::decode_kernel<1>), grid, blk, 0, st, d_in, bx, (int32_t)n, d_out, d_stride, d_modes, width, height); break;
    case 3: hipLaunchKernelGGL((itw::decode_kernel<3>), grid, blk, 0, st, d_in, bx, (int32_t)n, d_out, d_stride, d_modes, width, height); break;
    case 7: hipLaunchKernelGGL((itw::decode_kernel<7>), grid, blk, 0, st, d_in, bx, (int32_t)n, d_out, d_stride, d_modes, width, height); break;
    case 4: hipLaunchKernelGGL((itw::decode_kernel<4>), grid, blk, 0, st, d_in, bx, (int32_t)n, d_out, d_stride, d_modes, width, height); break;
    case 5: hipLaunchKernelGGL((itw::decode_kernel<5>), grid, blk, 0, st, d_in, bx, (int32_t)n, d_out, d_stride, d_modes, width, height); break;
    default: hipLaunchKernelGGL((itw::decode_kernel<6>), grid, blk, 0, st, d_in, bx, (int32_t)n, d_out, d_stride, d_modes, width, height); break;
    }
    DEC_CHECK(hipGetLastError());
    if (!dout) DEC_CHECK(hipMemcpy2DAsync(out, (size_t)out_stride, d_out, row_bytes, row_bytes, (size_t)height, hipMemcpyDeviceToHost, st));
    if (modes && !dmodes) DEC_CHECK(hipMemcpyAsync(modes, d_modes, (size_t)n * 4, hipMemcpyDeviceToHost, st));
    if (!din || !dout || !dmodes) {
        DEC_CHECK(hipStreamSynchronize(st));
        if (!din) (void)hipFree(d_in);
        if (!dout) (void)hipFree(d_out);
        if (modes && !dmodes) (void)hipFree(d_modes);
    }
    return 0;
}
