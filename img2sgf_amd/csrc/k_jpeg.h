// Device half of the JPEG decoder (host half: jpeg_host.h): dequantisation + inverse DCT of every
// 8x8 block, then chroma upsampling + colour conversion straight into an interleaved RGB image -- the pixels
// PIL.Image.open(path).convert("RGB") (img2sgf.py:651) yields, bit for bit.  Restated from libjpeg(-turbo), which Pillow
// runs with its defaults:
//   jidctint.c  jpeg_idct_islow  : 13-bit fixed-point LL&M inverse DCT, two passes (descale 11 and 18 bits), range limit
//   jdsample.c  h2v1 / h2v2 fancy upsampling ("triangle filter": 3/4 nearer + 1/4 further sample, alternating rounding)
//   jdcolor.c   ycc_rgb_convert  : 16-bit fixed-point tables
// Pinned against Pillow itself (tests/test_gpu_jpeg.py, tests/test_emu_pipeline.py).
#pragma once
#include "i2s_types.h"
#include "jpeg_host.h"

namespace i2s {

struct JpgDesc {
    const int16_t* coef[3];     // per component: bh * bw blocks of 64 coefficients, natural order
    uint8_t* plane[3];          // per component: (8 bh) x (8 bw) samples
    uint8_t* out;               // interleaved RGB, out_stride bytes per row
    int bw[3], bh[3], dw[3], dh[3];
    int ncomp, X, Y, out_stride;
    int hs, vs;                 // luma sampling factors (1 or 2): chroma is upsampled by these
    int nblocks[3];             // bw * bh
    unsigned short q[3][64];    // quantisation tables of the components, natural order
};

// ---- entropy decoding on the device, serial form: one lane per file runs the decoder of jpeg_host.h (the same
// source, compiled for the device) over the file's scans in order.  A lane needs ~3 700 cycles per symbol (three or four
// dependent global loads each), 0.8 s for a 220 KB file: 5-70 x slower than the host threads on passes of 16-256 files
// (profiles/r02_e_jpeg_entropy_rate_lanes.txt).  It is kept for the files the parallel decoder (k_jpeg_entropy.h) does not
// take -- progressive ones -- when everything is to stay on the device (i2s_params.jpeg_entropy_device = 2).
struct JpgHuffScan {
    int ns, ci[3], td[3], ta[3], ss, se, ah, al, dri;
    int tab_dc[4], tab_ac[4];     // indices into the pass's table pool (-1: not defined at this scan)
    unsigned off, len;            // entropy-coded segment inside the pass's byte blob
};
struct JpgHuffImg {
    JpegFrameView f;
    int scan0, nscans;
    int slot;                     // where the verdict goes
    int16_t* coef[3];
};

// grid (ceil(nb / 64)), block 64.  status[imgs[i].slot] = JPG_OK / JPG_BAD.
__global__ __launch_bounds__(64) void k_jpeg_huffman(const JpgHuffImg* __restrict__ imgs, const JpgHuffScan* __restrict__ scans,
                                                     const JpegHuff* __restrict__ tabs, const uint8_t* __restrict__ bytes, int nb,
                                                     int* __restrict__ status)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= nb) return;
    const JpgHuffImg& im = imgs[i];
    int16_t* coef[3] = {im.coef[0], im.coef[1], im.coef[2]};
    int rc = JPG_OK;
    for (int s = 0; s < im.nscans && rc == JPG_OK; s++) {
        const JpgHuffScan& hs = scans[im.scan0 + s];
        JpegScanView v;
        v.ns = hs.ns; v.ss = hs.ss; v.se = hs.se; v.ah = hs.ah; v.al = hs.al; v.dri = hs.dri;
        for (int k = 0; k < 3; k++) { v.ci[k] = hs.ci[k]; v.td[k] = hs.td[k]; v.ta[k] = hs.ta[k]; }
        for (int t = 0; t < 4; t++) {
            v.dc[t] = tabs + (hs.tab_dc[t] < 0 ? 0 : hs.tab_dc[t]);
            v.ac[t] = tabs + (hs.tab_ac[t] < 0 ? 0 : hs.tab_ac[t]);
        }
        v.data = bytes + hs.off; v.len = hs.len;
        rc = jpg_decode_scan_view(im.f, v, coef);
    }
    status[im.slot] = rc;
}

// One pass of the LL&M inverse DCT over 8 values; SHIFT = 11 (columns) or 18 (rows).
template <int SHIFT>
__device__ __forceinline__ void jpg_idct8(const int (&in)[8], int (&out)[8])
{
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * -15137, tmp3 = z1 + z2 * 6270;
    z2 = in[0]; z3 = in[4];
    int tmp0 = (z2 + z3) << 13, tmp1 = (z2 - z3) << 13;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    constexpr int RND = 1 << (SHIFT - 1);
    out[0] = (tmp10 + tmp3 + RND) >> SHIFT; out[7] = (tmp10 - tmp3 + RND) >> SHIFT;
    out[1] = (tmp11 + tmp2 + RND) >> SHIFT; out[6] = (tmp11 - tmp2 + RND) >> SHIFT;
    out[2] = (tmp12 + tmp1 + RND) >> SHIFT; out[5] = (tmp12 - tmp1 + RND) >> SHIFT;
    out[3] = (tmp13 + tmp0 + RND) >> SHIFT; out[4] = (tmp13 - tmp0 + RND) >> SHIFT;
}

// libjpeg's range-limit table around CENTERJSAMPLE, indexed with (x & 1023)
__device__ __forceinline__ int jpg_range_limit(int x)
{
    x &= 1023;
    return x < 128 ? x + 128 : (x < 512 ? 255 : (x < 896 ? 0 : x - 896));
}

// grid (ceil(max blocks per image / 64), nb), block 64: one 8x8 block per thread (a JPEG of a megapixel holds ~25 000 blocks).
__global__ __launch_bounds__(64) void k_jpeg_idct(const JpgDesc* __restrict__ jd)
{
    const JpgDesc& J = jd[blockIdx.y];
    int i = blockIdx.x * 64 + threadIdx.x, c = 0;
    while (c < J.ncomp && i >= J.nblocks[c]) { i -= J.nblocks[c]; c++; }
    if (c >= J.ncomp) return;
    const int16_t* cf = J.coef[c] + (size_t)i * 64;
    int ws[8][8];
#pragma unroll
    for (int col = 0; col < 8; col++) {
        int in[8], o[8];
#pragma unroll
        for (int r = 0; r < 8; r++) in[r] = (int)cf[r * 8 + col] * (int)J.q[c][r * 8 + col];
        jpg_idct8<11>(in, o);
#pragma unroll
        for (int r = 0; r < 8; r++) ws[r][col] = o[r];
    }
    const int by = i / J.bw[c], bx = i - by * J.bw[c];
    uint8_t* dst = J.plane[c] + ((size_t)by * 8) * (size_t)(J.bw[c] * 8) + (size_t)bx * 8;
#pragma unroll
    for (int r = 0; r < 8; r++) {
        int o[8];
        jpg_idct8<18>(ws[r], o);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { lo |= (unsigned)jpg_range_limit(o[k]) << (8 * k); hi |= (unsigned)jpg_range_limit(o[4 + k]) << (8 * k); }
        *reinterpret_cast<uint2*>(dst + (size_t)r * (size_t)(J.bw[c] * 8)) = make_uint2(lo, hi);
    }
}

// chroma sample for output pixel (x, y) of a component subsampled by (hs, vs) in {(1,1), (2,1), (2,2)}
__device__ __forceinline__ int jpg_chroma(const uint8_t* __restrict__ p, int pitch, int dw, int dh, int hs, int vs, int x, int y)
{
    if (hs == 1) return p[(size_t)y * pitch + x];
    const int c = x >> 1;
    // libjpeg only installs the fancy upsamplers for components more than 2 samples wide; narrower ones are replicated
    if (dw <= 2) return p[(size_t)(vs == 2 ? y >> 1 : y) * pitch + c];
    if (vs == 1) {
        // h2v1 fancy: 3/4 this + 1/4 neighbour, rounding 1 (left half) / 2 (right half); the outermost samples are copied
        const uint8_t* r = p + (size_t)y * pitch;
        const int v = r[c];
        if (x & 1) return c == dw - 1 ? v : (v * 3 + r[c + 1] + 2) >> 2;
        return c == 0 ? v : (v * 3 + r[c - 1] + 1) >> 2;
    }
    // h2v2 fancy: column sums 3 * nearer row + further row, then the same 3 : 1 rule across columns with rounding 8 / 7
    const int inrow = y >> 1;
    const int other = (y & 1) ? imin(inrow + 1, dh - 1) : imax(inrow - 1, 0);
    const uint8_t* r0 = p + (size_t)inrow * pitch;
    const uint8_t* r1 = p + (size_t)other * pitch;
    const int cs = r0[c] * 3 + r1[c];
    if (x & 1) return c == dw - 1 ? (cs * 4 + 7) >> 4 : (cs * 3 + r0[c + 1] * 3 + r1[c + 1] + 7) >> 4;
    return c == 0 ? (cs * 4 + 8) >> 4 : (cs * 3 + r0[c - 1] * 3 + r1[c - 1] + 8) >> 4;
}

// grid (ceil(w_max / 64), ceil(h_max / 4), nb), block (64, 4): one output pixel per thread.
__global__ __launch_bounds__(256) void k_jpeg_rgb(const JpgDesc* __restrict__ jd)
{
    const JpgDesc& J = jd[blockIdx.z];
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= J.X || y >= J.Y) return;
    uint8_t* o = J.out + (size_t)y * J.out_stride + (size_t)x * 3;
    const int yy = J.plane[0][(size_t)y * (size_t)(J.bw[0] * 8) + x];
    if (J.ncomp == 1) { o[0] = o[1] = o[2] = (uint8_t)yy; return; }       // Image.convert("RGB") of an "L" image
    const int cb = jpg_chroma(J.plane[1], J.bw[1] * 8, J.dw[1], J.dh[1], J.hs, J.vs, x, y) - 128;
    const int cr = jpg_chroma(J.plane[2], J.bw[2] * 8, J.dw[2], J.dh[2], J.hs, J.vs, x, y) - 128;
    // jdcolor.c build_ycc_rgb_table: FIX(1.40200) = 91881, FIX(1.77200) = 116130, FIX(0.71414) = 46802, FIX(0.34414) = 22554
    const int r = yy + ((91881 * cr + 32768) >> 16);
    const int g = yy + ((-22554 * cb + 32768 - 46802 * cr) >> 16);
    const int b = yy + ((116130 * cb + 32768) >> 16);
    o[0] = (uint8_t)iclamp(r, 0, 255); o[1] = (uint8_t)iclamp(g, 0, 255); o[2] = (uint8_t)iclamp(b, 0, 255);
}

}  // namespace i2s
