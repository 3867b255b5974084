// Pre-processing of the source image on the device, bit-exact with the reference's Pillow calls.
//
// Rotate + crop (crop_and_rotate_image, img2sgf.py:110-114): Image.rotate(angle, NEAREST, fillcolor="white", center=c)
// followed by Image.crop(box).  Pillow's nearest-neighbour affine transform (Geometry.c, affine_fixed) walks the output in
// 16.16 fixed point: with the inverse matrix (a0 a1 a2; a3 a4 a5) in fixed point, a2 / a5 taken at the pixel centre,
//   xin = (a2 + y * a1 + x * a0) >> 16,  yin = (a5 + y * a4 + x * a3) >> 16;  inside the source -> copy, else the fill colour.
// crop() then takes box (l, t, r, b) of that image; parts of the box outside it read 0.  The host computes the matrix
// exactly as Image.rotate() does (img2sgf_amd/preprocess.py) and the fixed-point constants exactly as Geometry.c does
// (i2s_api.hip: xform_fixed); the kernel writes the cropped region into the context's staging buffer.
//
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

struct XfDesc {
    const uint8_t* src;     // source image (host upload in the raw staging buffer, or caller's device pointer)
    int sstride, sw, sh;    // source row stride in bytes, width, height
    int a0, a1, a2, a3, a4, a5;   // 16.16 inverse affine matrix, a2 / a5 at the pixel centre
    int cl, ct;             // crop origin inside the rotated image
    int pad;
};

// grid (ceil(w_max / 64), ceil(h_max / 4), nb), block (64, 4): one output pixel per thread.  desc[b] already describes the
// OUTPUT (cropped) image in the staging buffer.
__global__ __launch_bounds__(256) void k_rotate_crop(const ImgDesc* __restrict__ desc, const XfDesc* __restrict__ xf)
{
    const int b = blockIdx.z;
    const ImgDesc im = desc[b];
    const XfDesc X = xf[b];
    const int ox = blockIdx.x * 64 + threadIdx.x, oy = blockIdx.y * 4 + threadIdx.y;
    if (ox >= im.w || oy >= im.h) return;
    uint8_t* dst = const_cast<uint8_t*>(im.src) + (size_t)oy * im.sstride + (size_t)ox * im.cn;
    const int x = ox + X.cl, y = oy + X.ct;              // position in the rotated image (same size as the source)
    const uint8_t* from = nullptr;
    int fill = 0;                                        // crop() outside the image
    if (x >= 0 && x < X.sw && y >= 0 && y < X.sh) {
        fill = 255;                                      // rotate()'s fillcolor "white"
        // Pillow accumulates xx += a0 per pixel and a2 += a1 per row in int: modular arithmetic
        const int xin = (int)((unsigned)X.a2 + (unsigned)y * (unsigned)X.a1 + (unsigned)x * (unsigned)X.a0) >> 16;
        const int yin = (int)((unsigned)X.a5 + (unsigned)y * (unsigned)X.a4 + (unsigned)x * (unsigned)X.a3) >> 16;
        if (xin >= 0 && xin < X.sw && yin >= 0 && yin < X.sh) from = X.src + (size_t)yin * X.sstride + (size_t)xin * im.cn;
    }
    for (int c = 0; c < im.cn; c++) dst[c] = from ? from[c] : (uint8_t)fill;
}

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
