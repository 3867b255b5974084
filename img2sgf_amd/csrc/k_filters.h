// Grey conversion and the blur bank (img2sgf.py:153, 171-175) as LDS-tiled integer kernels.
// All arithmetic is integer and bit-exact with OpenCV's 8-bit paths:
//   cvtColor BGR2GRAY   (color_rgb.simd.hpp, 15-bit coefficients)
//   GaussianBlur 8U     (smooth.dispatch.cpp / smooth.simd.hpp fixed-point 8.8 taps, REFLECT_101)
//   medianBlur 8U       (median_blur.simd.hpp, exact median, REPLICATE)
#pragma once
#include "i2s_types.h"
#include "tile_io.h"

namespace i2s {

constexpr int FT_W = 64;   // filter output tile
constexpr int FT_H = 32;

// ---- K1: grey plane.  cn==1: copy; cn==3: (ch0*B + ch1*G + ch2*R + half) >> shift, where the
// reference hands RGB data to COLOR_BGR2GRAY, so ch0 (=R) is weighted as "blue" (img2sgf.py:153).
// block (64,4), each thread 4 pixels.
__global__ __launch_bounds__(256) void k_grey(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ grey, int shift)
{
    const int b = blockIdx.z;
    const ImgDesc im = desc[b];
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (y >= im.h || x0 >= im.w) return;
    const uint8_t* s = im.src + (size_t)y * im.sstride;
    uint8_t* o = grey + (size_t)b * g.slot + (size_t)y * g.pitch;
    int cb, cg, cr;
    if (shift == 14) { cb = 1868; cg = 9617; cr = 4899; } else { cb = 3735; cg = 19235; cr = 9798; }
    for (int i = 0; i < 4; i++) {
        const int x = x0 + i;
        if (x >= im.w) break;
        if (im.cn == 1) o[x] = s[x];
        else o[x] = (uint8_t)((s[3 * x] * cb + s[3 * x + 1] * cg + s[3 * x + 2] * cr + (1 << (shift - 1))) >> shift);
    }
}

// ---- K3: separable fixed-point Gaussian, K in {3,5,7}.  4 pixels per thread, dword LDS traffic.
// horizontal: t = sum w_i * p (<= 65280, 16 bit); vertical: a = sum w_j * t (32 bit); out = (a + 32768) >> 16.
template <int K>
__global__ __launch_bounds__(256) void k_gauss(const ImgDesc* __restrict__ desc, Geo g,
                                               const uint8_t* __restrict__ grey, uint8_t* __restrict__ out, Taps taps)
{
    constexpr int R = K / 2;
    constexpr int SROWS = FT_H + 2 * R, SWORDS = FT_W / 4 + 2, SSTR = SWORDS + 1;   // bytes x0-4 .. x0+68
    constexpr int NS = FT_W / 4, HSTR = 2 * NS + 1;                                 // u16 pairs per row
    __shared__ unsigned s_src[SROWS * SSTR];
    __shared__ unsigned s_h[SROWS * HSTR];
    const int b = blockIdx.z;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    load_tile_words<SROWS, SWORDS, SSTR, 256, BORDER_R101>(s_src, grey + (size_t)b * g.slot, g.pitch, w, h, x0 - 4, y0 - R, tid);
    __syncthreads();
    for (int i = tid; i < SROWS * NS; i += 256) {
        const int ry = i / NS, s = i - ry * NS;
        const unsigned* ps = s_src + ry * SSTR + s;
        const unsigned wa = ps[0], wb = ps[1], wc = ps[2];
        int p[12];
#pragma unroll
        for (int q = 0; q < 4; q++) { p[q] = (int)((wa >> (8 * q)) & 0xffu); p[4 + q] = (int)((wb >> (8 * q)) & 0xffu); p[8 + q] = (int)((wc >> (8 * q)) & 0xffu); }
        unsigned t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            unsigned acc = 0;
#pragma unroll
            for (int j = 0; j < K; j++) acc += (unsigned)taps.k[j] * (unsigned)p[4 + q - R + j];
            t[q] = acc > 65535u ? 65535u : acc;
        }
        s_h[ry * HSTR + 2 * s] = t[0] | (t[1] << 16);
        s_h[ry * HSTR + 2 * s + 1] = t[2] | (t[3] << 16);
    }
    __syncthreads();
    uint8_t* o = out + (size_t)b * g.slot;
    for (int i = tid; i < FT_H * NS; i += 256) {
        const int ly = i / NS, s = i - ly * NS;
        unsigned a[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < K; j++) {
            const unsigned h0 = s_h[(ly + j) * HSTR + 2 * s], h1 = s_h[(ly + j) * HSTR + 2 * s + 1];
            const unsigned tj = (unsigned)taps.k[j];
            a[0] += tj * (h0 & 0xffffu); a[1] += tj * (h0 >> 16); a[2] += tj * (h1 & 0xffffu); a[3] += tj * (h1 >> 16);
        }
        unsigned ow = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const unsigned vv = (a[q] + 32768u) >> 16; ow |= (vv > 255u ? 255u : vv) << (8 * q); }
        const int x = x0 + 4 * s, y = y0 + ly;
        if (y < h && x < w) {
            uint8_t* dp = o + (size_t)y * g.pitch + x;
            if (x + 3 < w) *reinterpret_cast<unsigned*>(dp) = ow;
            else for (int q = 0; q < 4 && x + q < w; q++) dp[q] = (uint8_t)(ow >> (8 * q));
        }
    }
}

// ---- K4: exact KxK median, BORDER_REPLICATE.  The median m of n = K*K values is the largest t with
// #(values < t) <= n/2; built bit by bit (8 counting passes over the window held in registers).
// 4 pixels per thread; the (K + 3) x K neighbourhood of the strip is read as 3 dwords per row.
template <int K>
__global__ __launch_bounds__(256) void k_median(const ImgDesc* __restrict__ desc, Geo g,
                                                const uint8_t* __restrict__ grey, uint8_t* __restrict__ out)
{
    constexpr int R = K / 2, N = K * K, HALF = N / 2;
    constexpr int SROWS = FT_H + 2 * R, SWORDS = FT_W / 4 + 2, SSTR = SWORDS + 1;
    constexpr int NS = FT_W / 4;
    __shared__ unsigned s_src[SROWS * SSTR];
    const int b = blockIdx.z;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    load_tile_words<SROWS, SWORDS, SSTR, 256, BORDER_REPL>(s_src, grey + (size_t)b * g.slot, g.pitch, w, h, x0 - 4, y0 - R, tid);
    __syncthreads();
    uint8_t* o = out + (size_t)b * g.slot;
    for (int i = tid; i < FT_H * NS; i += 256) {
        const int ly = i / NS, s = i - ly * NS;
        const int x = x0 + 4 * s, y = y0 + ly;
        if (y >= h || x >= w) continue;
        int p[K][12];
#pragma unroll
        for (int j = 0; j < K; j++) {
            const unsigned* ps = s_src + (ly + j) * SSTR + s;
            const unsigned wa = ps[0], wb = ps[1], wc = ps[2];
#pragma unroll
            for (int q = 0; q < 4; q++) { p[j][q] = (int)((wa >> (8 * q)) & 0xffu); p[j][4 + q] = (int)((wb >> (8 * q)) & 0xffu); p[j][8 + q] = (int)((wc >> (8 * q)) & 0xffu); }
        }
        unsigned ow = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            int m = 0;
#pragma unroll
            for (int bit = 7; bit >= 0; bit--) {
                const int t = m | (1 << bit);
                int c = 0;
#pragma unroll
                for (int j = 0; j < K; j++)
#pragma unroll
                    for (int k = 0; k < K; k++) c += (p[j][4 + q - R + k] < t) ? 1 : 0;
                if (c <= HALF) m = t;
            }
            ow |= (unsigned)m << (8 * q);
        }
        uint8_t* dp = o + (size_t)y * g.pitch + x;
        if (x + 3 < w) *reinterpret_cast<unsigned*>(dp) = ow;
        else for (int q = 0; q < 4 && x + q < w; q++) dp[q] = (uint8_t)(ow >> (8 * q));
    }
}

}  // namespace i2s
