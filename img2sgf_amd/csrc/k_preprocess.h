// Contrast / brightness enhancement of the source image on the device, bit-exact with the reference's
// Pillow calls (img2sgf.py:141-149):
//   ImageEnhance.Contrast(img).enhance(fc):   degenerate = grey level int(mean(L) + 0.5), L = (R*19595 + G*38470 + B*7471 + 0x8000) >> 16
//   ImageEnhance.Brightness(img).enhance(fb): degenerate = 0
//   Image.blend(degenerate, img, f) (Pillow Blend.c): f == 0 -> degenerate, f == 1 -> img,
//       t = (float)d + f * (float)(p - d)   in float32;   0 <= f <= 1: (uint8) t;   else clip: t <= 0 -> 0, t >= 255 -> 255, else (uint8) t
// The image is modified in place in the context's staging buffer (never in caller memory).
#pragma once
#include "i2s_types.h"

namespace i2s {

// grid (ceil(h_max / 8), nb), block 256: sum of the luma of 8 rows per workgroup -> lsum[b] (exact integer).
__global__ __launch_bounds__(256) void k_luma_sum(const ImgDesc* __restrict__ desc, unsigned long long* __restrict__ lsum)
{
    __shared__ unsigned long long s_part[4];
    const int b = blockIdx.y;
    const ImgDesc im = desc[b];
    const int y0 = blockIdx.x * 8;
    if (y0 >= im.h) return;
    unsigned long long acc = 0;
    const int rows = imin(8, im.h - y0);
    for (int i = threadIdx.x; i < rows * im.w; i += 256) {
        const int y = y0 + i / im.w, x = i % im.w;
        const uint8_t* p = im.src + (size_t)y * im.sstride + (size_t)x * im.cn;
        if (im.cn == 1) acc += p[0];
        else acc += (unsigned)((p[0] * 19595u + p[1] * 38470u + p[2] * 7471u + 0x8000u) >> 16);
    }
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d);
    if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&lsum[b], s_part[0] + s_part[1] + s_part[2] + s_part[3]);
}

__device__ __forceinline__ int pil_blend(int d, int p, float f, int mode /* 0 copy img, 1 degenerate, 2 interpolate, 3 extrapolate */)
{
    if (mode == 0) return p;
    if (mode == 1) return d;
    const float t = (float)d + f * (float)(p - d);
    if (mode == 2) return (int)t & 0xff;
    return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}

__device__ __host__ inline int pil_blend_mode(float f) { return f == 1.0f ? 0 : (f == 0.0f ? 1 : ((f >= 0.0f && f <= 1.0f) ? 2 : 3)); }

// grid (ceil(w_max * 3 / 1024), h_max, nb), block 256: 4 bytes per thread, in place.
__global__ __launch_bounds__(256) void k_enhance(const ImgDesc* __restrict__ desc, const unsigned long long* __restrict__ lsum,
                                                 float fc, float fb)
{
    const int b = blockIdx.z;
    const ImgDesc im = desc[b];
    const int y = blockIdx.y;
    if (y >= im.h) return;
    const int rowb = im.w * im.cn;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= rowb) return;
    const int mean = (int)((double)lsum[b] / (double)((long long)im.w * im.h) + 0.5);
    const int mc = pil_blend_mode(fc), mb = pil_blend_mode(fb);
    uint8_t* p = const_cast<uint8_t*>(im.src) + (size_t)y * im.sstride;
    for (int i = 0; i < 4 && x0 + i < rowb; i++) {
        int v = p[x0 + i];
        v = pil_blend(mean, v, fc, mc);
        v = pil_blend(0, v, fb, mb);
        p[x0 + i] = (uint8_t)v;
    }
}

}  // namespace i2s
