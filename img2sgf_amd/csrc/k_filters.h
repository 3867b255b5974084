// Grey conversion and the blur bank (img2sgf.py:153, 171-175) as LDS-tiled integer kernels.
// All arithmetic is integer and bit-exact with OpenCV's 8-bit paths:
//   cvtColor BGR2GRAY   (color_rgb.simd.hpp, 15-bit coefficients)
//   GaussianBlur 8U     (smooth.dispatch.cpp / smooth.simd.hpp fixed-point 8.8 taps, REFLECT_101)
//   medianBlur 8U       (median_blur.simd.hpp, exact median, REPLICATE)
#pragma once
#include "i2s_types.h"

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

// ---- K3: separable fixed-point Gaussian, K in {3,5,7}.
// horizontal: t = sum w_i * p (<= 65280, 16 bit); vertical: a = sum w_j * t (32 bit); out = (a + 32768) >> 16.
template <int K>
__global__ __launch_bounds__(256) void k_gauss(const ImgDesc* __restrict__ desc, Geo g,
                                               const uint8_t* __restrict__ grey, uint8_t* __restrict__ out, Taps taps)
{
    constexpr int R = K / 2, LW = FT_W + 2 * R, LH = FT_H + 2 * R;
    __shared__ uint8_t s_in[LH][LW + 2];
    __shared__ uint16_t s_h[LH][FT_W];
    const int b = blockIdx.z;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    const uint8_t* src = grey + (size_t)b * g.slot;
    for (int i = tid; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = reflect101(y0 + ly - R, h), gx = reflect101(x0 + lx - R, w);
        s_in[ly][lx] = src[(size_t)gy * g.pitch + gx];
    }
    __syncthreads();
    for (int i = tid; i < LH * FT_W; i += 256) {
        const int ly = i / FT_W, lx = i - ly * FT_W;
        unsigned t = 0;
#pragma unroll
        for (int j = 0; j < K; j++) t += (unsigned)taps.k[j] * s_in[ly][lx + j];
        s_h[ly][lx] = (uint16_t)(t > 65535u ? 65535u : t);
    }
    __syncthreads();
    uint8_t* o = out + (size_t)b * g.slot;
    for (int i = tid; i < FT_H * FT_W; i += 256) {
        const int ly = i / FT_W, lx = i - ly * FT_W;
        unsigned a = 0;
#pragma unroll
        for (int j = 0; j < K; j++) a += (unsigned)taps.k[j] * s_h[ly + j][lx];
        const unsigned v = (a + 32768u) >> 16;
        const int x = x0 + lx, y = y0 + ly;
        if (x < w && y < h) o[(size_t)y * g.pitch + x] = (uint8_t)(v > 255u ? 255u : v);
    }
}

// ---- K4: exact KxK median, BORDER_REPLICATE.  The median m of n = K*K values is the largest t with
// #(values < t) <= n/2; built bit by bit (8 counting passes over the window held in registers).
template <int K>
__global__ __launch_bounds__(256) void k_median(const ImgDesc* __restrict__ desc, Geo g,
                                                const uint8_t* __restrict__ grey, uint8_t* __restrict__ out)
{
    constexpr int R = K / 2, LW = FT_W + 2 * R, LH = FT_H + 2 * R, N = K * K, HALF = N / 2;
    __shared__ uint8_t s_in[LH][LW + 2];
    const int b = blockIdx.z;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = blockIdx.x * FT_W, y0 = blockIdx.y * FT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    const uint8_t* src = grey + (size_t)b * g.slot;
    for (int i = tid; i < LH * LW; i += 256) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = iclamp(y0 + ly - R, 0, h - 1), gx = iclamp(x0 + lx - R, 0, w - 1);
        s_in[ly][lx] = src[(size_t)gy * g.pitch + gx];
    }
    __syncthreads();
    uint8_t* o = out + (size_t)b * g.slot;
    for (int i = tid; i < FT_H * FT_W; i += 256) {
        const int ly = i / FT_W, lx = i - ly * FT_W;
        int v[N];
#pragma unroll
        for (int j = 0; j < K; j++)
#pragma unroll
            for (int k = 0; k < K; k++) v[j * K + k] = s_in[ly + j][lx + k];
        int m = 0;
#pragma unroll
        for (int bit = 7; bit >= 0; bit--) {
            const int t = m | (1 << bit);
            int c = 0;
#pragma unroll
            for (int q = 0; q < N; q++) c += (v[q] < t) ? 1 : 0;
            if (c <= HALF) m = t;
        }
        const int x = x0 + lx, y = y0 + ly;
        if (x < w && y < h) o[(size_t)y * g.pitch + x] = (uint8_t)m;
    }
}

}  // namespace i2s
