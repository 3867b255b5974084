// Grey conversion and the blur bank (img2sgf.py:153, 171-175) as LDS-tiled integer kernels.
// All arithmetic is integer and bit-exact with OpenCV's 8-bit paths:
//   cvtColor BGR2GRAY   (color_rgb.simd.hpp, 15-bit coefficients)
//   GaussianBlur 8U     (smooth.dispatch.cpp / smooth.simd.hpp fixed-point 8.8 taps, REFLECT_101)
//   medianBlur 8U       (median_blur.simd.hpp, exact median, REPLICATE)
#pragma once
#include <type_traits>
#include "i2s_types.h"
#include "tile_io.h"
#include <gfx950_ops.h>

namespace i2s {

constexpr int FT_W = 64;   // filter output tile
constexpr int FT_H = 32;

constexpr int BL_R = 64;          // output rows per wavefront (+6 apron rows of horizontal work)
// bands of the two-valued speculation (k_blur; k_median57_bin for the integer-kernel path): a flag per 256 x 64 band
constexpr int MB_R = BL_R;        // output rows per wavefront (+6 apron rows); a band = 256 x MB_R pixels
__device__ __host__ inline int mb_bands_x(int wmax) { return (wmax + 255) / 256; }
__device__ __host__ inline int mb_bands_y(int hmax) { return (hmax + MB_R - 1) / MB_R; }

// ---- K1: grey plane.  cn==1: copy; cn==3: (ch0*B + ch1*G + ch2*R + half) >> shift, where the
// reference hands RGB data to COLOR_BGR2GRAY, so ch0 (=R) is weighted as "blue" (img2sgf.py:153).
// block (64,4), each thread 4 pixels of GREY_ROWS / 4 rows (a workgroup of 256 x 4 pixels, as in rounds 1-3, is 450 000 workgroups for a
// pass of 288 scans: the kernel was bound by their dispatch, 1.9 us per megapixel).
constexpr int GREY_ROWS = 32;
// has_colour[b] (zeroed by the host) is raised when a 3-channel image holds a pixel whose channels differ.  The reference hands EVERY input
// over as RGB (Image.open(...).convert("RGB"), img2sgf.py:651), also the greyscale scans that are most of its inputs: with R = G = B the
// grey value is that value under either set of weights and the 3-channel Canny picks channel 0 of three identical gradients, so such an
// image goes through the single-channel kernels (k_sobel_nms_rows on its grey plane) bit for bit.
// band_flags (zeroed by the host; the flags of k_blur's two-valued speculation, see there): raised for the 256 x BL_R band of this workgroup
// when a grey value other than 0 / 255 comes by -- scans are 97 % such bands, and neither k_blur nor the main Canny then starts a
// speculative walk that its first rows would stop.
__global__ __launch_bounds__(256) void k_grey(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ grey, int shift, int* __restrict__ has_colour,
                                              int* __restrict__ band_flags, int gx, int gy)
{
    const TileId t = tile_of_block(gx, gy);
    const int b = t.z;
    const ImgDesc im = desc[b];
    if (im.grey == im.src) return;                                   // the source is the grey plane (see ImgDesc)
    const int x0 = (t.tx * 64 + threadIdx.x) * 4;
    if (x0 >= im.w) return;
    int cb, cg, cr;
    if (shift == 14) { cb = 1868; cg = 9617; cr = 4899; } else { cb = 3735; cg = 19235; cr = 9798; }
    const int half = 1 << (shift - 1);
    bool coloured = false;
    unsigned odd = 0;                                                // bits 0 .. 6 of a byte: the pixel is neither 0 nor 255
    for (int y = t.ty * GREY_ROWS + threadIdx.y; y < imin((t.ty + 1) * GREY_ROWS, im.h); y += 4) {
    const uint8_t* s = im.src + (size_t)y * im.sstride;
    uint8_t* o = grey + (size_t)b * g.slot + rowoff(y, g.pitch);
    if (x0 + 3 < im.w) {
        // whole dwords; the source may start anywhere (odd-width RGB rows): the loads are unaligned dword loads
        unsigned out;
        if (im.cn == 1) __builtin_memcpy(&out, s + x0, 4);
        else {
            unsigned d0, d1, d2;                                          // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
            __builtin_memcpy(&d0, s + 3 * x0, 4); __builtin_memcpy(&d1, s + 3 * x0 + 4, 4); __builtin_memcpy(&d2, s + 3 * x0 + 8, 4);
            const unsigned g0 = ((d0 & 0xffu) * cb + ((d0 >> 8) & 0xffu) * cg + ((d0 >> 16) & 0xffu) * cr + half) >> shift;
            const unsigned g1 = ((d0 >> 24) * cb + (d1 & 0xffu) * cg + ((d1 >> 8) & 0xffu) * cr + half) >> shift;
            const unsigned g2 = (((d1 >> 16) & 0xffu) * cb + (d1 >> 24) * cg + (d2 & 0xffu) * cr + half) >> shift;
            const unsigned g3 = (((d2 >> 8) & 0xffu) * cb + ((d2 >> 16) & 0xffu) * cg + (d2 >> 24) * cr + half) >> shift;
            out = g0 | (g1 << 8) | (g2 << 16) | (g3 << 24);
            // R == G == B for each of the four pixels?
            const bool differs = ((d0 & 0xffu) != ((d0 >> 8) & 0xffu)) | (((d0 >> 8) & 0xffu) != ((d0 >> 16) & 0xffu)) |
                                 ((d0 >> 24) != (d1 & 0xffu)) | ((d1 & 0xffu) != ((d1 >> 8) & 0xffu)) |
                                 (((d1 >> 16) & 0xffu) != (d1 >> 24)) | ((d1 >> 24) != (d2 & 0xffu)) |
                                 (((d2 >> 8) & 0xffu) != ((d2 >> 16) & 0xffu)) | (((d2 >> 16) & 0xffu) != (d2 >> 24));
            if (differs) coloured = true;
        }
        *reinterpret_cast<unsigned*>(o + x0) = out;
        odd |= (out >> 1) ^ out;
    } else
    for (int i = 0; i < 4; i++) {
        const int x = x0 + i;
        if (x >= im.w) break;
        if (im.cn == 1) o[x] = s[x];
        else {
            o[x] = (uint8_t)((s[3 * x] * cb + s[3 * x + 1] * cg + s[3 * x + 2] * cr + half) >> shift);
            if (s[3 * x] != s[3 * x + 1] || s[3 * x + 1] != s[3 * x + 2]) coloured = true;
        }
        odd |= ((unsigned)o[x] >> 1) ^ o[x];
    }
    }
    if (coloured && has_colour[b] == 0) has_colour[b] = 1;          // (a plain store: every writer writes 1)
    if ((odd & 0x7f7f7f7fu) != 0u && band_flags != nullptr) {
        static_assert(BL_R % GREY_ROWS == 0, "a k_grey workgroup lies inside one band");
        int* f = band_flags + ((size_t)b * mb_bands_y(g.hmax) + t.ty * GREY_ROWS / BL_R) * mb_bands_x(g.wmax) + t.tx;
        if (*f == 0) *f = 1;
    }
}

// The channels of COLOURED 3-channel images as three planes (rgb + c * nb * slot, c = 0 .. 2, plane pitch): the input of the colour mode
// of k_sobel_nms_rows.  Images whose channels are equal everywhere (has_colour == 0, k_grey) are skipped.  block (64,4), 4 pixels per thread.
__global__ __launch_bounds__(256) void k_split_rgb(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ rgb, const int* __restrict__ has_colour,
                                                   int gx, int gy)
{
    const TileId t = tile_of_block(gx, gy);
    const int b = t.z;
    const ImgDesc im = desc[b];
    if (im.cn != 3 || has_colour[b] == 0) return;
    const int x0 = (t.tx * 64 + threadIdx.x) * 4;
    if (x0 >= im.w) return;
    for (int y = t.ty * GREY_ROWS + threadIdx.y; y < imin((t.ty + 1) * GREY_ROWS, im.h); y += 4) {
    const uint8_t* s = im.src + (size_t)y * im.sstride;
    unsigned o[3] = {0u, 0u, 0u};
    if (x0 + 3 < im.w) {
        unsigned d0, d1, d2;                                          // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3 (unaligned dword loads)
        __builtin_memcpy(&d0, s + 3 * x0, 4); __builtin_memcpy(&d1, s + 3 * x0 + 4, 4); __builtin_memcpy(&d2, s + 3 * x0 + 8, 4);
        o[0] = (d0 & 0xffu) | ((d0 >> 24) << 8) | (((d1 >> 16) & 0xffu) << 16) | (((d2 >> 8) & 0xffu) << 24);
        o[1] = ((d0 >> 8) & 0xffu) | ((d1 & 0xffu) << 8) | ((d1 >> 24) << 16) | (((d2 >> 16) & 0xffu) << 24);
        o[2] = ((d0 >> 16) & 0xffu) | (((d1 >> 8) & 0xffu) << 8) | ((d2 & 0xffu) << 16) | ((d2 >> 24) << 24);
    } else {
        for (int i = 0; i < 4 && x0 + i < im.w; i++)
#pragma unroll
            for (int c = 0; c < 3; c++) o[c] |= (unsigned)s[3 * (x0 + i) + c] << (8 * i);
    }
#pragma unroll
    for (int c = 0; c < 3; c++)                                       // (planes have a 64-byte pitch: whole dwords may be written)
        *reinterpret_cast<unsigned*>(rgb + ((size_t)c * g.nb + b) * g.slot + rowoff(y, g.pitch) + x0) = o[c];
    }
}

// ---- K3: separable fixed-point Gaussians 3x3 (sigma 3), 5x5 (sigma 5), 7x7 (sigma 7) of the blur bank, one kernel:
// the tile (+3 apron, BORDER_REFLECT_101) is fetched once and the three filters run back to back on it.
// 4 pixels per thread, dword LDS traffic.
// horizontal: t = sum w_i * p (<= 65280, 16 bit); vertical: a = sum w_j * t (32 bit); out = (a + 32768) >> 16.
constexpr int GA_R = 3;                                                     // apron of the widest filter
constexpr int GA_ROWS = FT_H + 2 * GA_R, GA_WORDS = FT_W / 4 + 2, GA_SSTR = GA_WORDS + 1;   // bytes x0-4 .. x0+68
constexpr int GA_NS = FT_W / 4, GA_HSTR = 2 * GA_NS + 1;                    // u16 pairs per row

template <int K>
__device__ __forceinline__ void gauss_on_tile(const unsigned* __restrict__ s_src, unsigned* __restrict__ s_h, const Taps& taps,
                                              uint8_t* __restrict__ o, int pitch, int w, int h, int x0, int y0, int tid)
{
    constexpr int R = K / 2, ROWS = FT_H + 2 * R, ROW0 = GA_R - R;          // tile rows ROW0 .. ROW0 + ROWS - 1 are needed
    for (int i = tid; i < ROWS * GA_NS; i += 256) {
        const int ry = i / GA_NS, s = i - ry * GA_NS;
        const unsigned* ps = s_src + (ROW0 + ry) * GA_SSTR + s;
        const unsigned wa = ps[0], wb = ps[1], wc = ps[2];
        int p[12];
#pragma unroll
        for (int q = 0; q < 4; q++) { p[q] = (int)((wa >> (8 * q)) & 0xffu); p[4 + q] = (int)((wb >> (8 * q)) & 0xffu); p[8 + q] = (int)((wc >> (8 * q)) & 0xffu); }
        unsigned t[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            unsigned acc = 0;
#pragma unroll
            for (int j = 0; j < K; j++) acc += __umul24((unsigned)taps.k[j], (unsigned)p[4 + q - R + j]);   // 24-bit multiply: full rate
            t[q] = acc > 65535u ? 65535u : acc;
        }
        s_h[ry * GA_HSTR + 2 * s] = t[0] | (t[1] << 16);
        s_h[ry * GA_HSTR + 2 * s + 1] = t[2] | (t[3] << 16);
    }
    __syncthreads();
    for (int i = tid; i < FT_H * GA_NS; i += 256) {
        const int ly = i / GA_NS, s = i - ly * GA_NS;
        unsigned a[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < K; j++) {
            const unsigned h0 = s_h[(ly + j) * GA_HSTR + 2 * s], h1 = s_h[(ly + j) * GA_HSTR + 2 * s + 1];
            const unsigned tj = (unsigned)taps.k[j];
            a[0] += __umul24(tj, h0 & 0xffffu); a[1] += __umul24(tj, h0 >> 16);       // taps <= 256, sums <= 65535
            a[2] += __umul24(tj, h1 & 0xffffu); a[3] += __umul24(tj, h1 >> 16);
        }
        unsigned ow = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) { const unsigned vv = (a[q] + 32768u) >> 16; ow |= (vv > 255u ? 255u : vv) << (8 * q); }
        const int x = x0 + 4 * s, y = y0 + ly;
        if (y < h && x < w) {
            uint8_t* dp = o + rowoff(y, pitch) + x;
            if (x + 3 < w) *reinterpret_cast<unsigned*>(dp) = ow;
            else for (int q = 0; q < 4 && x + q < w; q++) dp[q] = (uint8_t)(ow >> (8 * q));
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_gauss357(const ImgDesc* __restrict__ desc, Geo g,
                                                  uint8_t* __restrict__ out3, uint8_t* __restrict__ out5, uint8_t* __restrict__ out7,
                                                  Taps t3, Taps t5, Taps t7, int gx, int gy)
{
    __shared__ unsigned s_src[GA_ROWS * GA_SSTR];
    __shared__ unsigned s_h[GA_ROWS * GA_HSTR];
    const TileId t = tile_of_block(gx, gy);
    const int b = t.z;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = t.tx * FT_W, y0 = t.ty * FT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    load_tile_words<GA_ROWS, GA_WORDS, GA_SSTR, 256, BORDER_R101>(s_src, desc[b].grey, desc[b].gpitch, w, h, x0 - 4, y0 - GA_R, tid);
    __syncthreads();
    const size_t off = (size_t)b * g.slot;
    gauss_on_tile<3>(s_src, s_h, t3, out3 + off, g.pitch, w, h, x0, y0, tid);
    gauss_on_tile<5>(s_src, s_h, t5, out5 + off, g.pitch, w, h, x0, y0, tid);
    gauss_on_tile<7>(s_src, s_h, t7, out7 + off, g.pitch, w, h, x0, y0, tid);
}

// ---- K3 + K4(3x3) fused: the three Gaussians AND the 3x3 median of the blur bank in one pass over the grey plane, in
// registers -- no LDS, no barrier.  A lane owns one dword column (4 pixels) and walks down BL_R output rows; a wavefront
// is 64 consecutive dwords = 256 pixels of a row, the left / right neighbour dwords come from the adjacent lanes (lanes 0
// and 63 load theirs).  All arithmetic is FLOAT: measured on MI355X (profiles/r02_a_valu_rate_*.txt) v_add/mul/fma_f32 issue
// at the full rate while every integer multiply, min/max, bit-field or packed instruction takes twice as long, and the
// Gaussians are exact in f32: taps sum to 256 per axis, so every partial sum is an integer <= 255 * 256 * 256 < 2^24, and
// (a + 32768) >> 16 = trunc(a * 2^-16 + 0.5) with both operations exact (the sum has at most 24 significant bits); the
// 2^-16 is folded into the taps as 2^-8 per pass, which only moves exponents.
//   horizontal: pair sums p[-k] + p[k] shared by the three kernels, then 2 / 3 / 4 multiply-adds;
//   vertical:   the horizontal results of the last 7 rows live in a register ring (statically indexed: the row loop is
//               unrolled by 7), same pairing;
//   3x3 median: min3 / med3 / max3 of every column, then med3(max of minima, med of medians, min of maxima) (integers).
// Borders (REFLECT_101 for the Gaussians, REPLICATE for the median) are byte permutations of the (left, mid, right)
// dword triple with per-lane selectors computed once; only wavefronts that touch the left / right image edge execute them.
// Top / bottom: the Gaussians read row reflect101(y); the median ring repeats the first / last image row.
// Host side: used when every tap set sums to 256 (always for OpenCV's bit-exact kernels; the plain-rounding compatibility
// mode can give 257, for which the integer kernels below remain).
struct BlurTaps { float c3, a3, c5, a5, b5, c7, a7, b7, d7; };   // centre, +-1, +-2, +-3 of the 3 / 5 / 7-tap kernels
#define bl_f(v, byte) bl_fb<byte>(v)
// Machine-level pieces (isa/gfx950_ops.h): imin3 / imed3 / imax3, bl_vgpr + bl_fb (taps in vector registers, bytes converted
// with v_cvt_f32_ubyteN), bl_from_prev_lane / bl_from_next_lane (DPP whole-wave shifts), BlBuf loads / stores through a buffer
// descriptor, BL_CONSUME / BL_SCHED_FENCE (the wait for the prefetched row is placed BEFORE the row's stores).
// BORDER_REFLECT_101 of a row index that moves by one per step: (row, direction) instead of a modulo per row
struct BlReflect {
    int y, dir, n;
    __device__ __forceinline__ void init(int p, int n_) { n = n_; y = reflect101(p, n_); dir = 1;
        if (n_ > 1) { const int per = 2 * n_ - 2; int q = p % per; if (q < 0) q += per; dir = q < n_ - 1 ? 1 : -1; if (q == 0) dir = 1; } }
    __device__ __forceinline__ void step() { if (n > 1) { y += dir; if (y == n - 1) dir = -1; else if (y == 0) dir = 1; } }
};
__device__ __forceinline__ unsigned bl_pack(float a, float b, float c, float d)
{
    return (unsigned)a | ((unsigned)b << 8) | ((unsigned)c << 16) | ((unsigned)d << 24);
}
// (acc + 32768) >> 16 of the exact integer acc held in a float
__device__ __forceinline__ float bl_round16(float acc16) { return acc16 + 0.5f; }   // acc16 = acc * 2^-16 (exact); bl_pack truncates

// 3x3 medians of 4 adjacent pixels from three rows of 6 pixels (x0-1 .. x0+4), packed into one dword.  Integers: the
// three-operand min / med / max instructions cost the same for floats and ints, and float min / max would first have to
// canonicalise every input that crossed a basic block (IEEE mode), 18 more instructions per row.
__device__ __forceinline__ unsigned bl_median_row(const int (&A)[6], const int (&B)[6], const int (&Cc)[6])
{
    int lo[6], mi[6], hi[6];
#pragma unroll
    for (int i = 0; i < 6; i++) {
        lo[i] = imin3(A[i], B[i], Cc[i]);
        mi[i] = imed3(A[i], B[i], Cc[i]);
        hi[i] = imax3(A[i], B[i], Cc[i]);
    }
    unsigned o = 0;
#pragma unroll
    for (int q = 0; q < 4; q++)
        o |= (unsigned)imed3(imax3(lo[q], lo[q + 1], lo[q + 2]), imed3(mi[q], mi[q + 1], mi[q + 2]), imin3(hi[q], hi[q + 1], hi[q + 2])) << (8 * q);
    return o;
}

// the six median pixels x0-1 .. x0+4 (BORDER_REPLICATE) of a row from its raw dword triple
__device__ __forceinline__ void bl_median_pixels(unsigned L, unsigned M, unsigned R, bool fix, const unsigned (&mS)[3], int (&F)[6])
{
    unsigned ml = L, mm = M, mr = R;
    if (fix) { ml = __builtin_amdgcn_perm(M, L, mS[0]); mm = __builtin_amdgcn_perm(M, L, mS[1]); mr = __builtin_amdgcn_perm(R, M, mS[2]); }
    F[0] = (int)(ml >> 24);
    F[1] = (int)(mm & 0xffu); F[2] = (int)((mm >> 8) & 0xffu); F[3] = (int)((mm >> 16) & 0xffu); F[4] = (int)(mm >> 24);
    F[5] = (int)(mr & 0xffu);
}

#ifndef BL_WAVES
#define BL_WAVES 3
#endif
#ifndef BL_WAVES_BIN
#define BL_WAVES_BIN 4        // the two-valued kernel keeps its rings as 16-bit pairs: half the registers of the float rings
#endif
#ifndef BL_DEPTH
#define BL_DEPTH 6        // rows of loads in flight (3: 2.18, 4: 2.04, 5 / 6: 1.93, 8: 1.91 us per diagram)
#endif
// Two-valued bands.  A diagram of pure black and white (the benchmark's; a scan after the reference's contrast step mostly) needs no
// median machinery: with every pixel of the window 0 or 255 the 3x3 / 5x5 / 7x7 median is 255 iff at least 5 of 9 / 13 of 25 / 25
// of 49 are -- box sums and a compare, separable.  k_blur SPECULATES that its band (256 x BL_R pixels plus three rows and a dword on
// every side: everything the 7x7 window reaches) is such a band and then writes all six planes of the blur bank itself: per row the
// pixels become 0 / 1 bytes, the horizontal 3-, 5- and 7-sums are byte-shifted adds of the (left, mid, right) dword triple (no
// carries: a byte never exceeds 49), packed 2 + 3 + 3 bits per byte into ONE ring register per row; three running sums follow the
// rows, and sum + (128 - need) has its top bit set exactly when the majority is 255 (v_perm's sign selectors turn it into the byte).
// Every dword the band reads is tested for a byte that is neither 0 nor 255 (a byte is one of the two iff each bit equals the
// next higher one); at the first such byte the wavefront raises the band's flag and starts over in the general mode: Gaussians + the
// 3x3 sorting network, and k_median57 (the bit-serial kernel, below) then computes the 5x5 / 7x7 medians of every tile that touches a
// flagged band, overwriting whatever the speculation had written.  Either way every output pixel is exact.
// flags[(b * bands_y + band row) * bands_x + column group] != 0: the band holds a pixel other than 0 / 255 (zeroed by the host)
template <bool BIN_KERNEL>
__global__ __launch_bounds__(256, BIN_KERNEL ? BL_WAVES_BIN : BL_WAVES) void k_blur(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ med3,
                                                 uint8_t* __restrict__ out3, uint8_t* __restrict__ out5, uint8_t* __restrict__ out7,
                                                 uint8_t* __restrict__ med5, uint8_t* __restrict__ med7,
                                                 BlurTaps tps, int* __restrict__ band_flags, int gx, int gy)
{
    // taps scaled by 2^-8 (exact), used in both passes: the vertical sums then are acc * 2^-16 without a final multiply
    BlurTaps tp;
    constexpr float S = 1.0f / 256.0f;
    tp.c3 = bl_vgpr(tps.c3 * S); tp.a3 = bl_vgpr(tps.a3 * S);
    tp.c5 = bl_vgpr(tps.c5 * S); tp.a5 = bl_vgpr(tps.a5 * S); tp.b5 = bl_vgpr(tps.b5 * S);
    tp.c7 = bl_vgpr(tps.c7 * S); tp.a7 = bl_vgpr(tps.a7 * S); tp.b7 = bl_vgpr(tps.b7 * S); tp.d7 = bl_vgpr(tps.d7 * S);
    // the two-valued kernel: integer taps, one per 16-bit half (in vector registers for the same reason)
    auto pkt = [](float t) { const unsigned u = (unsigned)t; return bl_vgpr_u(u | (u << 16)); };
    const unsigned k_c3 = pkt(tps.c3), k_a3 = pkt(tps.a3), k_c5 = pkt(tps.c5), k_a5 = pkt(tps.a5), k_b5 = pkt(tps.b5);
    const unsigned k_c7 = pkt(tps.c7), k_a7 = pkt(tps.a7), k_b7 = pkt(tps.b7), k_d7 = pkt(tps.d7);
    // block = 4 wavefronts = 4 consecutive 256-pixel column groups of one band of BL_R rows
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z;
    const ImgDesc im = desc[b];
    const int w = im.w, h = im.h;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cgp = tl.tx * 4 + wave;                                // 256-pixel column group
    const int x0 = (cgp * 64 + lane) * 4;
    const int y0 = tl.ty * BL_R;
    if (cgp * 256 >= w || y0 >= h) return;                            // whole wavefront outside the image
    const bool active = x0 < w;
    const uint8_t* src = im.grey;
    const int sp = im.gpitch;
    const size_t obase = (size_t)b * g.slot;

    // per-lane border selectors: byte k of the (L, M, R) triple is pixel x0 - 4 + k
    const bool fix_lane = active && (x0 - 3 < 0 || x0 + 6 >= w);
    const bool fix = __any(fix_lane ? 1 : 0) != 0;
    // Gaussian: L' and M' only ever draw on (L, M) -- a reflection never reaches further than the own dword's neighbour --
    // R' may need all three dwords (two permutes).  Median: L', M' from (L, M), R' from (M, R).
    unsigned gA[3] = {0x03020100u, 0x07060504u, 0x0c0c0c0cu}, gB2 = 0x07060504u;
    unsigned mS[3] = {0x03020100u, 0x07060504u, 0x07060504u};
    if (fix_lane) {
#pragma unroll
        for (int d = 0; d < 3; d++) {
            unsigned a = 0, bsel = 0, m = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int p = x0 - 4 + 4 * d + j;
                int sg = reflect101(p, w) - (x0 - 4), sm = iclamp(p, 0, w - 1) - (x0 - 4);
                if (sg < 0 || sg > 11) sg = 4;                        // a pixel no valid output of this lane needs
                if (sm < 0 || sm > 11) sm = 4;
                a |= (unsigned)(sg < 8 ? sg : (d < 2 ? 4 : 0x0c)) << (8 * j);
                bsel |= (unsigned)(sg < 8 ? j : sg - 4) << (8 * j);
                // median: sources of L' and M' lie in (L, M), sources of R' in (M, R)
                const int ms = d < 2 ? sm : sm - 4;
                m |= (unsigned)(ms < 0 || ms > 7 ? 4 : ms) << (8 * j);
            }
            gA[d] = a; mS[d] = m;
            if (d == 2) gB2 = bsel;
        }
    }

    // one dword of a row for this lane, plus the edge dword of lanes 0 / 63 (their outer neighbour).  The loads are
    // UNCONDITIONAL (lanes without a valid column read offset 0 of the row): a predicated load keeps the old register value
    // for the other lanes, which makes the compiler wait for the previous load before it issues the next one.
    const bool has_e = (lane == 0 && x0 >= 4) || (lane == 63 && x0 + 4 < w);
    const unsigned xm = active ? (unsigned)x0 : 0u, xe = has_e ? (unsigned)(lane == 0 ? x0 - 4 : x0 + 4) : 0u;
    // bits 0 .. 6 of the bytes of the lane's dwords that are pixels of the image (only those are tested for "neither 0 nor 255")
    unsigned vm = 0, ve = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) if (active && x0 + q < w) vm |= 0x7fu << (8 * q);
    if (has_e) {
        const int xe0 = lane == 0 ? x0 - 4 : x0 + 4;
#pragma unroll
        for (int q = 0; q < 4; q++) if (xe0 + q < w) ve |= 0x7fu << (8 * q);
    }
    const BlBuf sbuf = bl_buf(src);
    const BlBuf o_m = bl_buf(med3 + obase), o_3 = bl_buf(out3 + obase), o_5 = bl_buf(out5 + obase), o_7 = bl_buf(out7 + obase);
    const BlBuf o_m5 = bl_buf(med5 + obase), o_m7 = bl_buf(med7 + obase);
    static_assert((BL_R + 6) % 7 == 0, "the row loop is unrolled by the ring depth");
    const int t_end = imin(BL_R + 6, h + 3 - (y0 - 3));               // input rows beyond h + 2 feed no output of this band

    // the median's (left, mid, right) triple of a row: BORDER_REPLICATE along x
    auto med_triple = [&](unsigned L, unsigned M, unsigned R, unsigned& ml, unsigned& mm, unsigned& mr) {
        ml = L; mm = M; mr = R;
        if (fix) { ml = __builtin_amdgcn_perm(M, L, mS[0]); mm = __builtin_amdgcn_perm(M, L, mS[1]); mr = __builtin_amdgcn_perm(R, M, mS[2]); }
    };
    // two-valued rows: the horizontal 3- / 5- / 7-sums of the 0 / 1 bytes, packed h3 | h5 << 2 | h7 << 5 per pixel byte
    auto bin_row = [&](unsigned L, unsigned M, unsigned R) -> unsigned {
        unsigned ml, mm, mr;
        med_triple(L, M, R, ml, mm, mr);
        const unsigned fl = ml & 0x01010101u, fm = mm & 0x01010101u, fr = mr & 0x01010101u;
        // byte q of the sums = pixel x0 + q: pixels x0 + q - 3 .. x0 + q + 3 are bytes q + 1 .. q + 7 of the triple
        const unsigned h3 = alignbyte(fm, fl, 3) + fm + alignbyte(fr, fm, 1);
        const unsigned h5 = h3 + alignbyte(fm, fl, 2) + alignbyte(fr, fm, 2);
        const unsigned h7 = h5 + alignbyte(fm, fl, 1) + alignbyte(fr, fm, 3);
        return h3 | (h5 << 2) | (h7 << 5);
    };

    // One walk down the band.  BIN: the speculative mode (returns false at the first byte that is neither 0 nor 255).
    auto walk = [&](auto bin_tag) -> bool {
        constexpr bool BIN = decltype(bin_tag)::value;
        float H3[7][4], H5[7][4], H7[7][4];        // horizontal results of the last 7 input rows, slot = row index mod 7
        unsigned P3[7][2], P5[7][2], P7[7][2];     // BIN: the same as 16-bit pairs ([0]: pixels x0, x0 + 2; [1]: x0 + 1, x0 + 3), values <= 256
        int F1[6], F2[6];                           // general mode, 3x3 median ring: pixels x0-1 .. x0+4 of the two previous rows
        unsigned RB[7];                             // BIN: packed row sums of the last 7 rows (BORDER_REPLICATE rows), slot = row index mod 7
        unsigned S3 = 0, S5 = 0, S7 = 0;            // BIN: vertical running sums (rows t-4 .. t-2, t-5 .. t-1, t-6 .. t)
        unsigned p_first = 0, p_last = 0;           // BIN: packed sums of image row 0 (bands at the top) / of the latest image row
#pragma unroll
        for (int i = 0; i < 7; i++) {
            RB[i] = 0;
            P3[i][0] = P3[i][1] = P5[i][0] = P5[i][1] = P7[i][0] = P7[i][1] = 0u;
#pragma unroll
            for (int q = 0; q < 4; q++) { H3[i][q] = 0.f; H5[i][q] = 0.f; H7[i][q] = 0.f; }
        }
#pragma unroll
        for (int i = 0; i < 6; i++) { F1[i] = 0; F2[i] = 0; }
        if (y0 == 0) {
            // BORDER_REPLICATE above the image: rows -3 .. -1 are row 0
            const unsigned M = bl_bload(sbuf, 0, xm), E = bl_bload(sbuf, 0, xe);
            const unsigned L = bl_from_prev_lane(M, E), R = bl_from_next_lane(M, E);
            if (BIN) {
                if (__any(((((M >> 1) ^ M) & vm) | (((E >> 1) ^ E) & ve)) != 0u)) return false;
                p_first = bin_row(L, M, R);
            } else bl_median_pixels(L, M, R, fix, mS, F1);
        }
        // BL_DEPTH rows in flight: vmcnt counts loads and stores together and in order, so the wait for a row's pixels is also a
        // wait for every store issued before that load -- the deeper the queue, the more rows of stores may still be on their way
        // The rows in flight sit in a RING of 7 slots, statically indexed like the result rings below (slot = row index mod 7): row t is
        // read from slot u, the load of row t + BL_DEPTH goes to slot (u + BL_DEPTH) % 7.  Rounds 2-3 kept them in a shift queue
        // (q[d] = q[d + 1] every row): the moves READ the newest entries, so the compiler had to wait for the load issued one row
        // earlier -- s_waitcnt vmcnt(10) -- and with it, the counter being in order, for every store but the last row's: at most eight
        // stores in flight per wavefront, 3.4 TB/s of plane writes where the same load / store scheme streams 6.2
        // (tools/micro/store_bw_bench.hip; ablations in profiles/r04_b_blur_experiments.txt).
        static_assert(BL_DEPTH >= 1 && BL_DEPTH <= 6, "the ring has 7 slots: at most 6 rows in flight beside the current one");
        BlReflect ry;
        ry.init(y0 - 3, h);
        unsigned qM[7], qE[7];
#pragma unroll
        for (int d = 0; d < 7; d++) { qM[d] = 0; qE[d] = 0; }
#pragma unroll
        for (int d = 0; d < BL_DEPTH; d++) {
            if (d > 0) ry.step();
            const int ro = __builtin_amdgcn_readfirstlane(rowoff(ry.y, sp));      // uniform: without this every load sits in a waterfall loop
            qM[d] = bl_bload(sbuf, ro, xm);
            qE[d] = bl_bload(sbuf, ro, xe);
            // The compiler's s_waitcnt for a ring slot must hold on the FIRST trip through the row loop too, where the slot was loaded
            // here: it counts the memory operations issued after that load on the shortest path.  The prologue therefore issues what a
            // row of the loop issues -- its stores, rejected by the range check -- so that the count is the loop's (BL_DEPTH - 1 rows of
            // loads AND stores may stay in flight) and not the prologue's loads alone (vmcnt(10): hardly more than one row of stores)
#pragma unroll
            for (int k = 0; k < (BIN ? 6 : 5); k++) bl_bstore(o_3, 0, BL_NO_STORE, 0u);
        }
        // (the first row is tested before the loop -- its load has to have arrived for the first trip anyway -- so that a band of a noisy
        // image is given up at once and not after seven rows)
        unsigned odd_acc = BIN ? ((((qM[0] >> 1) ^ qM[0]) & vm) | (((qE[0] >> 1) ^ qE[0]) & ve)) : 0u;
        for (int t0 = 0; t0 < t_end; t0 += 7) {
            // (the test for a byte that is neither 0 nor 255 leaves the row loop only here: an exit inside the unrolled rows makes the
            // compiler's wait counts fall back to the prologue's, see above; what the rows since the last test stored is overwritten)
            if (BIN && __any(odd_acc != 0u)) return false;
#pragma unroll
            for (int u = 0; u < 7; u++) {
                const int t = t0 + u;
                const int yi = y0 - 3 + t;
                const unsigned M = qM[u], E = qE[u];
                unsigned& nM = qM[(u + 1) % 7];
                unsigned& nE = qE[(u + 1) % 7];
                {
                    ry.step();
                    const int ro = __builtin_amdgcn_readfirstlane(rowoff(ry.y, sp));
                    qM[(u + BL_DEPTH) % 7] = bl_bload(sbuf, ro, xm);
                    qE[(u + BL_DEPTH) % 7] = bl_bload(sbuf, ro, xe);
                }
                if (BIN) odd_acc |= (((M >> 1) ^ M) & vm) | (((E >> 1) ^ E) & ve);      // tested once per 7 rows, below
                const unsigned L = bl_from_prev_lane(M, E), R = bl_from_next_lane(M, E);
                unsigned gl = L, gm = M, gr = R;
                if (fix) {
                    BL_KEEP_BRANCH();
                    gl = __builtin_amdgcn_perm(M, L, gA[0]);
                    gm = __builtin_amdgcn_perm(M, L, gA[1]);
                    gr = __builtin_amdgcn_perm(R, __builtin_amdgcn_perm(M, L, gA[2]), gB2);
                }
                // the row's 0 / 1 bytes and their shifted copies (BIN): byte q of A<k> / B<k> = pixel x0 + q - k / x0 + q + k
                unsigned bA1 = 0, bA2 = 0, bA3 = 0, bB1 = 0, bB2 = 0, bB3 = 0, bC = 0;
                if (BIN) {
                    // horizontal pass on the 0 / 1 BYTES (four pixels per add), weights as 16-bit pairs (two pixels per multiply-add):
                    // a pixel is 255 b, so the Gaussian is (255 v + 32768) >> 16 with v = sum of tap products over the window's ones --
                    // h = sum_k w_k b <= 256 here, v <= 65536 in the vertical pass
                    const unsigned fl = gl & 0x01010101u, fm = gm & 0x01010101u, fr = gr & 0x01010101u;
                    bA1 = alignbyte(fm, fl, 3); bB1 = alignbyte(fr, fm, 1);
                    bA2 = alignbyte(fm, fl, 2); bB2 = alignbyte(fr, fm, 2);
                    bA3 = alignbyte(fm, fl, 1); bB3 = alignbyte(fr, fm, 3);
                    bC = fm;
                    const unsigned s1 = bA1 + bB1, s2 = bA2 + bB2, s3 = bA3 + bB3;
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) {
                        const unsigned ce = (bC >> (8 * hf)) & 0x00ff00ffu, e1 = (s1 >> (8 * hf)) & 0x00ff00ffu;
                        const unsigned e2 = (s2 >> (8 * hf)) & 0x00ff00ffu, e3 = (s3 >> (8 * hf)) & 0x00ff00ffu;
                        P3[u][hf] = pk_mad_u16(k_a3, e1, pk_mul_u16(k_c3, ce));
                        P5[u][hf] = pk_mad_u16(k_b5, e2, pk_mad_u16(k_a5, e1, pk_mul_u16(k_c5, ce)));
                        P7[u][hf] = pk_mad_u16(k_d7, e3, pk_mad_u16(k_b7, e2, pk_mad_u16(k_a7, e1, pk_mul_u16(k_c7, ce))));
                    }
                } else {
                    float f[10];                                              // pixels x0 - 3 .. x0 + 6
                    f[0] = bl_f(gl, 1); f[1] = bl_f(gl, 2); f[2] = bl_f(gl, 3);
                    f[3] = bl_f(gm, 0); f[4] = bl_f(gm, 1); f[5] = bl_f(gm, 2); f[6] = bl_f(gm, 3);
                    f[7] = bl_f(gr, 0); f[8] = bl_f(gr, 1); f[9] = bl_f(gr, 2);
                    // horizontal pass of the three Gaussians into ring slot u
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const float c = f[q + 3];
                        const float s1 = f[q + 2] + f[q + 4], s2 = f[q + 1] + f[q + 5], s3 = f[q] + f[q + 6];
                        H3[u][q] = __builtin_fmaf(tp.a3, s1, tp.c3 * c);
                        H5[u][q] = __builtin_fmaf(tp.b5, s2, __builtin_fmaf(tp.a5, s1, tp.c5 * c));
                        H7[u][q] = __builtin_fmaf(tp.d7, s3, __builtin_fmaf(tp.b7, s2, __builtin_fmaf(tp.a7, s1, tp.c7 * c)));
                    }
                }
                const int yo = yi - 3;
                const bool st_g = t >= 6 && yo < h;
                unsigned om = 0, om2 = 0, om5 = 0, om7 = 0;
                bool st_m = false, st_m2 = false;
                if (BIN) {
                    // the medians' row is BORDER_REPLICATE: rows above the image are row 0, rows below it the last one (the loaded row
                    // is the Gaussians' REFLECT_101 row there)
                    unsigned p;
                    if (yi < 0) p = p_first;
                    else if (yi >= h) p = p_last;
                    else {
                        // (away from the image's left / right edge the medians' REPLICATE triple is the Gaussians' REFLECT_101 triple)
                        if (fix) p = bin_row(L, M, R);
                        else {
                            const unsigned h3 = bA1 + bC + bB1, h5 = h3 + bA2 + bB2, h7 = h5 + bA3 + bB3;
                            p = h3 | (h5 << 2) | (h7 << 5);
                        }
                        p_last = p;
                    }
                    const unsigned old = RB[u];                               // row t - 7
                    S7 += ((p >> 5) & 0x07070707u) - ((old >> 5) & 0x07070707u);
                    S5 += ((RB[(u + 6) % 7] >> 2) & 0x07070707u) - ((RB[(u + 1) % 7] >> 2) & 0x07070707u);
                    S3 += (RB[(u + 5) % 7] & 0x03030303u) - (RB[(u + 2) % 7] & 0x03030303u);
                    RB[u] = p;
                    om = bytes_from_sign(S3 + 0x7b7b7b7bu);                   // + (128 - 5): top bit <=> at least 5 of 9
                    om5 = bytes_from_sign(S5 + 0x73737373u);                  // + (128 - 13)
                    om7 = bytes_from_sign(S7 + 0x67676767u);                  // + (128 - 25)
                } else if (yi >= 0 && yi < h) {
                    // 3x3 median: image row yi completes output row yi - 1 (rows yi - 2, yi - 1, yi = F2, F1, F0); the last image
                    // row also completes itself (BORDER_REPLICATE below the image: rows h - 2, h - 1, h - 1)
                    int F0[6];
                    bl_median_pixels(L, M, R, fix, mS, F0);
                    const int ym = yi - 1;
                    if (ym >= y0 && ym < y0 + BL_R) { om = bl_median_row(F2, F1, F0); st_m = true; }
                    if (yi == h - 1 && yi >= y0 && yi < y0 + BL_R) { om2 = bl_median_row(F1, F0, F0); st_m2 = true; }
#pragma unroll
                    for (int i = 0; i < 6; i++) { F2[i] = F1[i]; F1[i] = F0[i]; }
                }
                // vertical pass: output row yo = yi - 3 from ring slots t-6 .. t (centre t-3)
                unsigned o3w = 0, o5w = 0, o7w = 0;
                if (st_g && BIN) {
                    constexpr int NS = 7;
                    const int c = (u + 4) % NS, p1 = (u + 5) % NS, m1 = (u + 3) % NS, p2 = (u + 6) % NS, m2 = (u + 2) % NS, p3 = u, m3 = (u + 1) % NS;
                    // v per pixel in 16 bits: every partial sum lacks a positive term of the all-ones window's 65536, only the last
                    // multiply-add can reach it -- that one saturates at 65535, and (255 * 65535 + 32768) >> 16 is 255 as well
                    unsigned v3[2], v5[2], v7[2];
#pragma unroll
                    for (int hf = 0; hf < 2; hf++) {
                        v3[hf] = pk_mad_u16_sat(k_c3, P3[c][hf], pk_mul_u16(k_a3, pk_add_u16(P3[p1][hf], P3[m1][hf])));
                        v5[hf] = pk_mad_u16_sat(k_c5, P5[c][hf], pk_mad_u16(k_a5, pk_add_u16(P5[p1][hf], P5[m1][hf]),
                                                                             pk_mul_u16(k_b5, pk_add_u16(P5[p2][hf], P5[m2][hf]))));
                        v7[hf] = pk_mad_u16_sat(k_c7, P7[c][hf], pk_mad_u16(k_a7, pk_add_u16(P7[p1][hf], P7[m1][hf]),
                                                pk_mad_u16(k_b7, pk_add_u16(P7[p2][hf], P7[m2][hf]), pk_mul_u16(k_d7, pk_add_u16(P7[p3][hf], P7[m3][hf])))));
                    }
                    // (255 v + 32768) >> 16: byte 2 of a 32-bit dot product per pixel, gathered with two byte permutes
                    auto finish = [](const unsigned (&v)[2]) {
                        const unsigned q0 = udot2_u16(v[0], 0x000000ffu, 32768u), q2 = udot2_u16(v[0], 0x00ff0000u, 32768u);
                        const unsigned q1 = udot2_u16(v[1], 0x000000ffu, 32768u), q3 = udot2_u16(v[1], 0x00ff0000u, 32768u);
                        return __builtin_amdgcn_perm(q1, q0, 0x0c0c0602u) | __builtin_amdgcn_perm(q3, q2, 0x06020c0cu);
                    };
                    o3w = finish(v3); o5w = finish(v5); o7w = finish(v7);
                }
                if (st_g && !BIN) {
                    constexpr int NS = 7;
                    const int c = (u + 4) % NS, p1 = (u + 5) % NS, m1 = (u + 3) % NS, p2 = (u + 6) % NS, m2 = (u + 2) % NS, p3 = u, m3 = (u + 1) % NS;
                    float r3[4], r5[4], r7[4];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        r3[q] = bl_round16(__builtin_fmaf(tp.a3, H3[p1][q] + H3[m1][q], tp.c3 * H3[c][q]));
                        r5[q] = bl_round16(__builtin_fmaf(tp.b5, H5[p2][q] + H5[m2][q],
                                                          __builtin_fmaf(tp.a5, H5[p1][q] + H5[m1][q], tp.c5 * H5[c][q])));
                        r7[q] = bl_round16(__builtin_fmaf(tp.d7, H7[p3][q] + H7[m3][q],
                                                          __builtin_fmaf(tp.b7, H7[p2][q] + H7[m2][q],
                                                                         __builtin_fmaf(tp.a7, H7[p1][q] + H7[m1][q], tp.c7 * H7[c][q]))));
                    }
                    o3w = bl_pack(r3[0], r3[1], r3[2], r3[3]);
                    o5w = bl_pack(r5[0], r5[1], r5[2], r5[3]);
                    o7w = bl_pack(r7[0], r7[1], r7[2], r7[3]);
                }
                // the stores of this row go out AFTER the wait for the next row's pixels (see BL_CONSUME)
                BL_SCHED_FENCE();
                BL_CONSUME(nM, nE);
                BL_SCHED_FENCE();
                // every store is issued on every path (see BL_NO_STORE): lanes beyond the image and rows that complete nothing get an
                // offset the buffer's range check rejects
                {
                    const int off = rowoff(yo, g.pitch);
                    const unsigned xg = (active && st_g) ? xm : BL_NO_STORE;
                    bl_bstore(o_3, off, xg, o3w);
                    bl_bstore(o_5, off, xg, o5w);
                    bl_bstore(o_7, off, xg, o7w);
                    if (BIN) {
                        bl_bstore(o_m, off, xg, om);
                        bl_bstore(o_m5, off, xg, om5);
                        bl_bstore(o_m7, off, xg, om7);
                    } else {
                        bl_bstore(o_m, rowoff(yi - 1, g.pitch), (active && st_m) ? xm : BL_NO_STORE, om);
                        bl_bstore(o_m, rowoff(yi, g.pitch), (active && st_m2) ? xm : BL_NO_STORE, om2);
                    }
                }
            }
        }
        return !(BIN && __any(odd_acc != 0u));
    };
    int* flag = band_flags ? band_flags + ((size_t)b * mb_bands_y(g.hmax) + tl.ty) * mb_bands_x(g.wmax) + cgp : nullptr;
    if (BIN_KERNEL) {
        // the speculative walk; a band that turns out not to be two-valued is flagged and left to the general kernel behind this one
        // (k_grey has flagged the bands in which it saw such a pixel already)
        if (*flag != 0) return;
        if (!walk(std::true_type{}) && lane == 0) *flag = 1;
    } else {
        if (flag != nullptr && *flag == 0) return;                    // the two-valued kernel has written all six planes of this band
        walk(std::false_type{});
    }
}

// ---- K4: exact medians, BORDER_REPLICATE (cv.medianBlur, img2sgf.py:174).

// 3x3: sort the three values of every column once (min3 / med3 / max3, shared by the three windows that contain the
// column), then median = med3( max of the column minima, med of the column medians, min of the column maxima ).
// 4 pixels per thread, dword LDS traffic.
__global__ __launch_bounds__(256) void k_median3(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ out, int gx, int gy)
{
    constexpr int SROWS = FT_H + 2, SWORDS = FT_W / 4 + 2, SSTR = SWORDS + 1;
    constexpr int NS = FT_W / 4;
    __shared__ unsigned s_src[SROWS * SSTR];
    const TileId t = tile_of_block(gx, gy);
    const int b = t.z;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = t.tx * FT_W, y0 = t.ty * FT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    load_tile_words<SROWS, SWORDS, SSTR, 256, BORDER_REPL>(s_src, desc[b].grey, desc[b].gpitch, w, h, x0 - 4, y0 - 1, tid);
    __syncthreads();
    uint8_t* o = out + (size_t)b * g.slot;
    for (int i = tid; i < FT_H * NS; i += 256) {
        const int ly = i / NS, s = i - ly * NS;
        const int x = x0 + 4 * s, y = y0 + ly;
        if (y >= h || x >= w) continue;
        int lo[6], mi[6], hi[6];
        {
            const unsigned* ps = s_src + ly * SSTR + s;
            int r[3][6];
#pragma unroll
            for (int j = 0; j < 3; j++) unpack6(ps[j * SSTR], ps[j * SSTR + 1], ps[j * SSTR + 2], r[j]);
#pragma unroll
            for (int c = 0; c < 6; c++) {
                lo[c] = imin3(r[0][c], r[1][c], r[2][c]);
                mi[c] = imed3(r[0][c], r[1][c], r[2][c]);
                hi[c] = imax3(r[0][c], r[1][c], r[2][c]);
            }
        }
        unsigned ow = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int a = imax3(lo[q], lo[q + 1], lo[q + 2]);
            const int m = imed3(mi[q], mi[q + 1], mi[q + 2]);
            const int c = imin3(hi[q], hi[q + 1], hi[q + 2]);
            ow |= (unsigned)imed3(a, m, c) << (8 * q);
        }
        uint8_t* dp = o + rowoff(y, g.pitch) + x;
        if (x + 3 < w) *reinterpret_cast<unsigned*>(dp) = ow;
        else for (int q = 0; q < 4 && x + q < w; q++) dp[q] = (uint8_t)(ow >> (8 * q));
    }
}

// ---- K4, two-valued bands: 5x5 and 7x7 medians of 0 / 255 pixels are majority votes ------------------------------------------
// A diagram of pure black and white (the benchmark's; a scan after the reference's contrast step mostly) needs no median
// machinery at all: with every pixel of the window 0 or 255, the 5x5 median is 255 iff at least 13 of the 25 are, the 7x7
// median iff at least 25 of the 49 -- a box sum and a compare, separable.  k_median57_bin runs first, register-resident in the
// manner of k_blur: a lane owns one dword column (4 pixels), a wavefront 256 pixels of a row, and walks down MB_R output
// rows; per input row the pixels become 0 / 1 bytes, the horizontal 5- and 7-sums are byte-shifted adds of the (left, mid,
// right) dword triple (no carries: a byte never exceeds 49), rings of the last 7 rows keep the vertical running sums, and
// sum + (128 - need) has its top bit set exactly when the majority is 255 (v_perm's sign selectors turn that bit into the byte).
// SPECULATIVE and exact: every dword the band reads (3 rows above and below, one dword left and right) is also tested for a
// byte that is neither 0 nor 255; at the first such byte the wavefront raises its band's flag and leaves.  k_median57 (the
// general bit-serial kernel, below) then runs on every tile that touches a flagged band and overwrites whatever the band
// wrote; tiles whose bands all stayed silent return at once.  Either way every output pixel is an exact median.

// flags[(b * bands_y + band row) * bands_x + column group] != 0: the band holds a pixel other than 0 / 255 (zeroed by the host)
__global__ __launch_bounds__(256) void k_median57_bin(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ out5,
                                                      uint8_t* __restrict__ out7, int* __restrict__ flags, int gx, int gy)
{
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z;
    const ImgDesc im = desc[b];
    const int w = im.w, h = im.h;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cgp = tl.tx * 4 + wave;                                // 256-pixel column group
    const int x0 = (cgp * 64 + lane) * 4;
    const int y0 = tl.ty * MB_R;
    if (cgp * 256 >= w || y0 >= h) return;                            // whole wavefront outside the image
    int* flag = flags + ((size_t)b * mb_bands_y(g.hmax) + tl.ty) * mb_bands_x(g.wmax) + cgp;
    const bool active = x0 < w;
    const size_t obase = (size_t)b * g.slot;
    // BORDER_REPLICATE along x: byte k of the (L, M, R) triple is pixel x0 - 4 + k; lanes whose 3 x 4 bytes reach outside the
    // image rebuild the triple with byte permutes (selectors computed once): L' = perm(M, L), M' = perm(M, L), R' = perm(R, M)
    const bool fix_lane = active && (x0 - 4 < 0 || x0 + 7 >= w);
    const bool fix = __any(fix_lane ? 1 : 0) != 0;
    unsigned sL = 0x03020100u, sM = 0x07060504u, sR = 0x07060504u;
    if (fix_lane) {
        sL = 0; sM = 0; sR = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            sL |= (unsigned)iclamp(iclamp(x0 - 4 + j, 0, w - 1) - (x0 - 4), 0, 7) << (8 * j);
            sM |= (unsigned)iclamp(iclamp(x0 + j, 0, w - 1) - (x0 - 4), 0, 7) << (8 * j);
            sR |= (unsigned)iclamp(iclamp(x0 + 4 + j, 0, w - 1) - x0, 0, 7) << (8 * j);
        }
    }
    // bytes of the lane's dwords that are pixels of the image (only those are tested for "neither 0 nor 255")
    unsigned vm = 0;
#pragma unroll
    for (int q = 0; q < 4; q++) if (x0 + q < w) vm |= 0x7fu << (8 * q);
    const bool has_e = (lane == 0 && x0 >= 4) || (lane == 63 && x0 + 4 < w);
    unsigned ve = 0;
    if (has_e) {
        const int xe0 = lane == 0 ? x0 - 4 : x0 + 4;
#pragma unroll
        for (int q = 0; q < 4; q++) if (xe0 + q < w) ve |= 0x7fu << (8 * q);
    }
    const unsigned xm = active ? (unsigned)x0 : 0u, xe = has_e ? (unsigned)(lane == 0 ? x0 - 4 : x0 + 4) : 0u;
    if (!active) vm = 0;

    unsigned H5[7], H7[7];                      // horizontal sums of the last 7 input rows, slot = row index mod 7
#pragma unroll
    for (int i = 0; i < 7; i++) { H5[i] = 0; H7[i] = 0; }
    unsigned S5 = 0, S7 = 0;                    // vertical running sums
    const BlBuf sbuf = bl_buf(im.grey);
    const int sp = im.gpitch;
    // two rows in flight (the kernel is bound by memory latency, not by its dozen instructions per pixel; 4, 6, 8 rows: no faster)
    unsigned nM, nE, n2M, n2E;
    {
        const int ro = rowoff(iclamp(y0 - 3, 0, h - 1), sp), ro2 = rowoff(iclamp(y0 - 2, 0, h - 1), sp);
        nM = bl_bload(sbuf, ro, xm);
        nE = bl_bload(sbuf, ro, xe);
        n2M = bl_bload(sbuf, ro2, xm);
        n2E = bl_bload(sbuf, ro2, xe);
    }
    static_assert((MB_R + 6) % 7 == 0, "the row loop is unrolled by the ring depth");
    const int t_end = imin(MB_R + 6, h + 3 - (y0 - 3));               // input rows beyond h + 2 feed no output of this band
    const BlBuf o_5 = bl_buf(out5 + obase), o_7 = bl_buf(out7 + obase);
    for (int t0 = 0; t0 < t_end; t0 += 7) {
#pragma unroll
        for (int u = 0; u < 7; u++) {
            const int t = t0 + u;
            const int yi = y0 - 3 + t;                                 // input row (BORDER_REPLICATE: clamped when outside)
            const unsigned M = nM, E = nE;
            nM = n2M; nE = n2E;
            {
                const int ro = rowoff(iclamp(yi + 2, 0, h - 1), sp);
                n2M = bl_bload(sbuf, ro, xm);
                n2E = bl_bload(sbuf, ro, xe);
            }
            // a byte is 0 or 255 iff each of its bits equals the next higher one
            const unsigned odd = (((M >> 1) ^ M) & vm) | (((E >> 1) ^ E) & ve);
            if (__any(odd != 0u)) {
                if (lane == 0) *flag = 1;
                return;
            }
            unsigned L = bl_from_prev_lane(M, E), R = bl_from_next_lane(M, E), Mf = M;
            if (fix) {
                const unsigned l2 = __builtin_amdgcn_perm(M, L, sL), m2 = __builtin_amdgcn_perm(M, L, sM), r2 = __builtin_amdgcn_perm(R, M, sR);
                L = l2; Mf = m2; R = r2;
            }
            const unsigned fl = L & 0x01010101u, fm = Mf & 0x01010101u, fr = R & 0x01010101u;
            // byte q of the sums = pixel x0 + q: pixels x0 + q - 2 .. x0 + q + 2 are bytes q + 2 .. q + 6 of the triple
            const unsigned h5 = alignbyte(fm, fl, 2) + alignbyte(fm, fl, 3) + fm + alignbyte(fr, fm, 1) + alignbyte(fr, fm, 2);
            const unsigned h7 = h5 + alignbyte(fm, fl, 1) + alignbyte(fr, fm, 3);
            // output row yo = yi - 3: 7x7 = input rows t - 6 .. t, 5x5 = rows t - 5 .. t - 1
            S7 += h7 - H7[u];
            S5 += H5[(u + 6) % 7] - H5[(u + 1) % 7];
            H7[u] = h7; H5[u] = h5;
            const int yo = yi - 3;
            const bool st = t >= 6 && yo < h;
            const unsigned o5 = bytes_from_sign(S5 + 0x73737373u);    // + (128 - 13): top bit <=> at least 13 of 25
            const unsigned o7 = bytes_from_sign(S7 + 0x67676767u);    // + (128 - 25): top bit <=> at least 25 of 49
            BL_SCHED_FENCE();
            BL_CONSUME(nM, nE);
            BL_SCHED_FENCE();
            if (st && active) {
                const int off = rowoff(yo, g.pitch);
                bl_bstore(o_5, off, xm, o5);
                bl_bstore(o_7, off, xm, o7);
            }
        }
    }
}

// 5x5 and 7x7 together, exact for any input, BIT-SLICED (round 3; round 2's kernel kept a pixel's window as bits of one
// register and paid two popcounts and a dozen dependent bookkeeping instructions per pixel, window and bit plane: ~270 lane
// instructions per pixel).  Here a register holds ONE BIT OF 32 NEIGHBOURING PIXELS and every step is a bitwise instruction
// on all 32 at once -- v_and / v_xor / v_bitop3, the full-rate instructions of this chip (profiles/r02_a_valu_rate_*.txt).
// The tile is transposed once into 8 bit planes (one 64-bit word per plane and row, bit i = pixel x0 - 4 + i).  A thread
// owns 24 adjacent output pixels of one row (a 32-bit window of the row words: 24 + the aprons) and reads the median off MSB
// first, the classic radix selection, but for 24 pixels in parallel:
//   cand[dy][dx]  (K x K registers, bit e = element column e): is the element at offset (dy, dx) from the output pixel whose
//                 window starts at column e - dx still a candidate, i.e. does it agree with that pixel's median in all higher bits;
//   per plane:    t = cand & plane row (no shifts: everything is aligned on the ELEMENT's column); the K values of a column are
//                 counted with carry-save adders (a full adder is two v_bitop3: parity and majority), the K column counts are
//                 shifted onto the output's column and added by an adder tree -> the 6-bit number c of candidates with a 1;
//                 the median's bit is 1 iff c > m (m = rank still to be found - 1, itself bit-sliced): the borrow of m - c;
//                 where the bit is 0, m becomes m - c; cand &= ~(plane ^ bit).
// 7x7: 49 + 56 + 18 + 36 + 18 + 6 + 49 instructions per plane and 24 pixels, 5x5: 25 + 30 + 12 + 24 + 15 + 4 + 25 -- 17 per
// pixel and plane for both medians where round 2 spent 28, and full-rate ones.
// Content adaptivity as before (exact): a plane that equals the next higher plane over the whole tile gets no round -- after the
// upper plane's round all candidates agree in that bit, hence in this one, so the median's bit repeats.
constexpr int MT_W = 48, MT_H = 128;                 // outputs per tile: two 24-pixel halves per row, one row per thread
constexpr int M_ROWS = MT_H + 6, M_SSTR = 17;

// full adder / half adder on 32 bit-sliced lanes: parity and majority are one v_bitop3_b32 each
__device__ __forceinline__ void bs_fa(unsigned a, unsigned b, unsigned c, unsigned& s, unsigned& cy) { s = bitop3<0x96>(a, b, c); cy = bitop3<0xE8>(a, b, c); }
__device__ __forceinline__ void bs_ha(unsigned a, unsigned b, unsigned& s, unsigned& cy) { s = a ^ b; cy = a & b; }

// K 3-bit numbers (v0 = ones, v1 = twos, v2 = fours) -> their 6-bit sum, column by column
__device__ __forceinline__ void bs_sum7(const unsigned (&v0)[7], const unsigned (&v1)[7], const unsigned (&v2)[7], unsigned (&c)[6])
{
    unsigned s, t, u, k0, k1, k2, k3, k4, k5;
    // ones: 7 bits
    bs_fa(v0[0], v0[1], v0[2], s, k0); bs_fa(v0[3], v0[4], v0[5], t, k1); bs_fa(s, t, v0[6], c[0], k2);
    // twos: 7 + 3
    unsigned a0, a1, a2, b0, b1, b2, b3;
    bs_fa(v1[0], v1[1], v1[2], a0, b0); bs_fa(v1[3], v1[4], v1[5], a1, b1); bs_fa(v1[6], k0, k1, a2, b2);
    bs_fa(a0, a1, a2, s, b3); bs_ha(s, k2, c[1], k3);
    // fours: 7 + 5 (b0 .. b3, k3)
    unsigned d0, d1, d2, d3, e0, e1, e2, e3, e4;
    bs_fa(v2[0], v2[1], v2[2], d0, e0); bs_fa(v2[3], v2[4], v2[5], d1, e1); bs_fa(v2[6], b0, b1, d2, e2); bs_fa(b2, b3, k3, d3, e3);
    bs_fa(d0, d1, d2, s, e4); bs_ha(s, d3, c[2], k4);
    // eights: 6 (e0 .. e4, k4)
    unsigned f0, f1, g0, g1;
    bs_fa(e0, e1, e2, f0, g0); bs_fa(e3, e4, k4, f1, g1); bs_ha(f0, f1, c[3], k5);
    // sixteens: 3
    bs_fa(g0, g1, k5, c[4], c[5]);
    (void)t; (void)u;
}
__device__ __forceinline__ void bs_sum5(const unsigned (&v0)[5], const unsigned (&v1)[5], const unsigned (&v2)[5], unsigned (&c)[6])
{
    unsigned s, k0, k1;
    // ones: 5
    bs_fa(v0[0], v0[1], v0[2], s, k0); bs_fa(s, v0[3], v0[4], c[0], k1);
    // twos: 5 + 2
    unsigned a0, a1, b0, b1, b2;
    bs_fa(v1[0], v1[1], v1[2], a0, b0); bs_fa(v1[3], v1[4], k0, a1, b1); bs_fa(a0, a1, k1, c[1], b2);
    // fours: 5 + 3
    unsigned d0, d1, e0, e1, e2, e3;
    bs_fa(v2[0], v2[1], v2[2], d0, e0); bs_fa(v2[3], v2[4], b0, d1, e1); bs_fa(d0, d1, b1, s, e2); bs_ha(s, b2, c[2], e3);
    // eights: 4
    unsigned f0, g0, g1;
    bs_fa(e0, e1, e2, f0, g0); bs_ha(f0, e3, c[3], g1);
    // sixteens: 2 (a sum of 25 needs no bit 5)
    c[4] = g0 ^ g1; c[5] = 0u;
}

// the K values of one element column: K one-bit inputs -> 3-bit count
__device__ __forceinline__ void bs_col7(const unsigned (&t)[7], unsigned& o0, unsigned& o1, unsigned& o2)
{
    unsigned s1, c1, s2, c2, c3;
    bs_fa(t[0], t[1], t[2], s1, c1); bs_fa(t[3], t[4], t[5], s2, c2); bs_fa(s1, s2, t[6], o0, c3); bs_fa(c1, c2, c3, o1, o2);
}
__device__ __forceinline__ void bs_col5(const unsigned (&t)[5], unsigned& o0, unsigned& o1, unsigned& o2)
{
    unsigned s1, c1, c2;
    bs_fa(t[0], t[1], t[2], s1, c1); bs_fa(s1, t[3], t[4], o0, c2); bs_ha(c1, c2, o1, o2);
}

// The K x K medians of 24 pixels of tile row `row` (output row index; its window rows are plane rows row + 3 - K/2 ..), whose
// 32-bit window of the plane row words starts at bit `sh` (0 or 24).  One round per LIVE plane (live_list: their numbers, most
// significant first, four bits each; a loop, not unrolled code: eight unrolled rounds of two medians are 40 KB of instructions);
// round i leaves its result -- bit (k - K/2) = that bit of the median of the pixel at window bit k, k = 4 .. 27 -- in
// s_res[i * 256 + thread].
// one round: plane p of the K window rows starting at `rows`; FIRST = no candidate has been excluded yet (plane 7's round),
// LAST = nobody will look at the candidates or the rank again
template <int K, bool FIRST, bool LAST>
__device__ __forceinline__ void bs_round(const unsigned long long* __restrict__ rows, int p, int sh, unsigned (&cand)[K][K], unsigned (&m)[6],
                                         unsigned* __restrict__ s_res_i)
{
    unsigned W[K];
#pragma unroll
    for (int dy = 0; dy < K; dy++) {
        const unsigned long long v = rows[p * M_ROWS + dy];
        W[dy] = __builtin_amdgcn_alignbit((unsigned)(v >> 32), (unsigned)v, (unsigned)sh);
    }
    unsigned v0[K], v1[K], v2[K];
#pragma unroll
    for (int dx = 0; dx < K; dx++) {
        unsigned t[K];
#pragma unroll
        for (int dy = 0; dy < K; dy++) t[dy] = FIRST ? W[dy] : (cand[dy][dx] & W[dy]);
        if (K == 7) bs_col7(reinterpret_cast<const unsigned (&)[7]>(t), v0[dx], v1[dx], v2[dx]);
        else bs_col5(reinterpret_cast<const unsigned (&)[5]>(t), v0[dx], v1[dx], v2[dx]);
        // onto the output's column: the element at offset dx of the window that starts at column a sits at column a + dx
        if (dx) { v0[dx] >>= dx; v1[dx] >>= dx; v2[dx] >>= dx; }
    }
    unsigned c[6];
    if (K == 7) bs_sum7(reinterpret_cast<const unsigned (&)[7]>(v0), reinterpret_cast<const unsigned (&)[7]>(v1), reinterpret_cast<const unsigned (&)[7]>(v2), c);
    else bs_sum5(reinterpret_cast<const unsigned (&)[5]>(v0), reinterpret_cast<const unsigned (&)[5]>(v1), reinterpret_cast<const unsigned (&)[5]>(v2), c);
    // m - c: the final borrow says c > m, i.e. at least `rank` candidates have a 1: the median's bit is 1
    unsigned dif[6], bor = 0u;
#pragma unroll
    for (int b = 0; b < 6; b++) {
        dif[b] = bitop3<0x96>(m[b], c[b], bor);
        bor = bitop3<0x8E>(m[b], c[b], bor);                                // (~m & (c | bor)) | (c & bor)
    }
    const unsigned ge = bor;
    *s_res_i = ge;
    if (LAST) return;
#pragma unroll
    for (int b = 0; b < 6; b++) m[b] = bitop3<0xCA>(ge, m[b], dif[b]);       // ge ? m : m - c -- bit 0: the c candidates with a 1 are larger than the median
#pragma unroll
    for (int dx = 0; dx < K; dx++) {
        const unsigned gsh = ge << dx;
#pragma unroll
        for (int dy = 0; dy < K; dy++)                                      // cand & ~(plane ^ bit)
            cand[dy][dx] = FIRST ? ~(W[dy] ^ gsh) : bitop3<0x90>(cand[dy][dx], W[dy], gsh);
    }
}

// The K x K medians of 24 pixels of tile row `row` (output row index; its window rows are plane rows row + 3 - K/2 ..), whose
// 32-bit window of the plane row words starts at bit `sh` (0 or 24).  One round per LIVE plane (live_list: their numbers, most
// significant first, four bits each; plane 7 is always the first, the others run in a loop, not as unrolled code: eight unrolled
// rounds of two medians are 40 KB of instructions); round i leaves its result -- bit (k - K/2) = that bit of the median of the
// pixel at window bit k, k = 4 .. 27 -- in s_res[i * 256 + thread].
template <int K>
__device__ __forceinline__ void bs_median_row(const unsigned long long* __restrict__ s_pl, int row, int sh, int nlive, unsigned live_list,
                                              unsigned* __restrict__ s_res)
{
    constexpr int R = K / 2;
    unsigned cand[K][K];
    unsigned m[6];                                     // rank still to be found, minus one, bit-sliced: starts at (K * K - 1) / 2
#pragma unroll
    for (int b = 0; b < 6; b++) m[b] = (((K * K - 1) / 2) >> b) & 1 ? 0xffffffffu : 0u;
    const unsigned long long* rows = s_pl + row + 3 - R;
    if (nlive == 1) { bs_round<K, true, true>(rows, 7, sh, cand, m, s_res + threadIdx.x); return; }
    bs_round<K, true, false>(rows, 7, sh, cand, m, s_res + threadIdx.x);
#pragma unroll 1
    for (int i = 1; i + 1 < nlive; i++)
        bs_round<K, false, false>(rows, (int)((live_list >> (4 * i)) & 7u), sh, cand, m, s_res + i * 256 + threadIdx.x);
    bs_round<K, false, true>(rows, (int)((live_list >> (4 * (nlive - 1))) & 7u), sh, cand, m, s_res + (nlive - 1) * 256 + threadIdx.x);
}

// the 8 result planes of a thread from the rounds' results: a plane without a round repeats the nearest live plane above it
__device__ __forceinline__ void bs_collect(const unsigned* __restrict__ s_res, unsigned live, int shift, unsigned (&res)[8])
{
#pragma unroll
    for (int p = 0; p < 8; p++) res[p] = s_res[(__popc(live >> p) - 1) * 256 + threadIdx.x] >> shift;
}

// 8 result planes (bit j = output pixel j, j < 24) -> 24 bytes = 6 dwords: per group of 8 pixels the plane bytes are gathered
// into a 64-bit word (byte p = plane p) and transposed as an 8 x 8 bit matrix (byte i = pixel i)
__device__ __forceinline__ void bs_planes_to_bytes(const unsigned (&pl)[8], unsigned (&out)[6])
{
#pragma unroll
    for (int gq = 0; gq < 3; gq++) {
        // selectors: byte gq of the low operand into byte 0 / 2, of the high operand into byte 1 / 3
        const unsigned sel01 = (unsigned)gq | ((unsigned)(gq + 4) << 8) | 0x0c0c0000u;
        const unsigned p01 = __builtin_amdgcn_perm(pl[1], pl[0], sel01), p23 = __builtin_amdgcn_perm(pl[3], pl[2], sel01);
        const unsigned p45 = __builtin_amdgcn_perm(pl[5], pl[4], sel01), p67 = __builtin_amdgcn_perm(pl[7], pl[6], sel01);
        unsigned long long x = (unsigned long long)__builtin_amdgcn_perm(p23, p01, 0x05040100u) |
                               ((unsigned long long)__builtin_amdgcn_perm(p67, p45, 0x05040100u) << 32);
        unsigned long long t;
        t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x = x ^ t ^ (t << 7);
        t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x = x ^ t ^ (t << 14);
        t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x = x ^ t ^ (t << 28);
        out[2 * gq] = (unsigned)x; out[2 * gq + 1] = (unsigned)(x >> 32);
    }
}

// A thread's 24 output bytes of one row.  The planes have a 64-byte pitch (whole dwords may be written) and xo is a multiple of 8: three
// 8-byte stores where the row has all 24 pixels (the wavefront's 64 lanes are 64 different ROWS, so every store instruction touches
// 64 cache lines: half as many instructions as dword stores, 3.84 -> 3.57 us per noisy diagram; without any store the kernel takes 3.08), dword stores at the image's right edge.
__device__ __forceinline__ void bs_store24(uint8_t* p, const unsigned (&o)[6], int xo, int w)
{
    if (xo + 20 < w) {
        uint2* q = reinterpret_cast<uint2*>(p);
        q[0] = make_uint2(o[0], o[1]); q[1] = make_uint2(o[2], o[3]); q[2] = make_uint2(o[4], o[5]);
    } else {
#pragma unroll
        for (int k = 0; k < 6; k++) if (xo + 4 * k < w) *reinterpret_cast<unsigned*>(p + 4 * k) = o[k];
    }
}

// The general kernel.  flags (optional): the band flags of k_median57_bin; a tile none of whose bands is flagged has exact
// medians already.  A workgroup looks at M_TPB consecutive tiles of the (plane, row-major tile) sequence: on two-valued
// diagrams all it does is read their flags, one tile per lane (one workgroup per tile, or one tile after the other, spent
// 0.3 us per diagram on nothing but load latencies).  The tiles it has to compute are pipelined: the source words of the next
// one are fetched into registers while the current one is computed.
constexpr int M_TPB = 8;
constexpr int M_WPT = (M_ROWS * 16 + 255) / 256;     // source dwords per thread and tile
#ifndef M_WAVES
#define M_WAVES 3
#endif
__global__ __launch_bounds__(256, M_WAVES) void k_median57(const ImgDesc* __restrict__ desc, Geo g,
                                                  uint8_t* __restrict__ out5, uint8_t* __restrict__ out7,
                                                  const int* __restrict__ flags, int gx, int gy, int ntiles)
{
    __shared__ unsigned s_src[M_ROWS * M_SSTR];      // source tile; during the rounds: their results (256 dwords per live plane)
    __shared__ unsigned long long s_pl[8 * M_ROWS];
    __shared__ unsigned s_differs;                   // bit p: plane p differs from plane p + 1 somewhere in the tile
    static_assert(M_ROWS * M_SSTR >= 8 * 256, "the rounds' results take the source tile's place");
    const unsigned first = tile_chunk_of_block(M_TPB, (unsigned)ntiles);
    const int tid = threadIdx.x;
    // lane k of every wavefront looks at tile first + k (the loads of all M_TPB tiles are in flight together); the wavefronts of
    // the workgroup compute the same mask
    const int lane = tid & 63;
    bool need = false;
    if (lane < M_TPB && first + lane < (unsigned)ntiles) {
        const TileId tl = tile_of_index(first + lane, gx, gy);
        const int b = tl.z;
        const int w = desc[b].w, h = desc[b].h;
        const int x0 = tl.tx * MT_W, y0 = tl.ty * MT_H;
        if (x0 < w && y0 < h) {
            need = flags == nullptr;
            if (flags) {
                // bands (256 x MB_R pixels) this tile's OUTPUT pixels lie in: at most 2 x 3 of them.  Three things make k_blur's speculation
                // exact: the flag array is indexed with the CONTEXT's band grid (g.wmax / g.hmax never change after i2s_create,
                // whatever the sizes of a pass's images), every tile whose output touches a flagged band is recomputed here, and the host
                // clears the flags at the start of every pass (run_pass) -- test_blur_flags_across_passes_of_different_sizes
                static_assert(MT_H <= 2 * MB_R && MT_W <= 256, "a tile's outputs lie in at most 2 x 3 bands");
                const int bx0 = x0 / 256, bx1 = imin(x0 + MT_W - 1, w - 1) / 256, by0 = y0 / MB_R, by1 = imin(y0 + MT_H - 1, h - 1) / MB_R;
                const int nbx = mb_bands_x(g.wmax), nby = mb_bands_y(g.hmax);
                int any = 0;
                for (int by = by0; by <= by1; by++)
                    for (int bx = bx0; bx <= bx1; bx++) any |= flags[((size_t)b * nby + by) * nbx + bx];
                need = any != 0;
            }
        }
    }
    unsigned long long todo = __ballot(need);
    if (!todo) return;
    // source words of a tile into registers (BORDER_REPLICATE)
    unsigned pre[M_WPT];
    auto fetch = [&](int k) {
        const TileId tl = tile_of_index(first + (unsigned)k, gx, gy);
        const ImgDesc im = desc[tl.z];
        const int xa = tl.tx * MT_W - 4, ya = tl.ty * MT_H - 3;
        if (xa >= 0 && ya >= 0 && xa + 64 <= im.w && ya + M_ROWS <= im.h) {
            // the tile and its apron lie inside the image (block-uniform): plain dword loads, no border arithmetic
            const uint8_t* base = im.grey + rowoff(ya, im.gpitch) + xa + 4 * (tid & 15);
#pragma unroll
            for (int q = 0; q < M_WPT; q++) {
                const int i = tid + q * 256;
                pre[q] = 0u;
                if (i < M_ROWS * 16) pre[q] = *reinterpret_cast<const unsigned*>(base + rowoff(i >> 4, im.gpitch));
            }
            return;
        }
#pragma unroll
        for (int q = 0; q < M_WPT; q++) {
            const int i = tid + q * 256;
            pre[q] = 0u;
            if (i < M_ROWS * 16) pre[q] = tile_word<BORDER_REPL>(im.grey, im.gpitch, im.w, im.h, xa + 4 * (i & 15), ya + (i >> 4));
        }
    };
    int k = __ffsll(todo) - 1;
    todo &= todo - 1ull;
    fetch(k);
    for (;;) {
        const TileId tl = tile_of_index(first + (unsigned)k, gx, gy);
        const int b = tl.z;
        const int w = desc[b].w, h = desc[b].h;
        const int x0 = tl.tx * MT_W, y0 = tl.ty * MT_H;
#pragma unroll
        for (int q = 0; q < M_WPT; q++) {
            const int i = tid + q * 256;
            if (i < M_ROWS * 16) s_src[(i >> 4) * M_SSTR + (i & 15)] = pre[q];
        }
        if (tid == 0) s_differs = 0;
        __syncthreads();                               // (also: everybody is done with the previous tile's planes and results)
        const int kn = todo ? __ffsll(todo) - 1 : -1;
        todo &= todo - 1ull;
        if (kn >= 0) fetch(kn);                        // in flight during this tile's rounds
        {
            // 8 pixels -> 8 plane bytes (8x8 bit-matrix transpose: output byte p = plane p, bit i = pixel i)
            uint8_t* plb = reinterpret_cast<uint8_t*>(s_pl);
            unsigned long long dacc = 0;                 // byte p: where plane p and plane p + 1 differ (over this thread's items)
            for (int i = tid; i < M_ROWS * 8; i += 256) {
                const int r = i >> 3, gq = i & 7;
                unsigned long long x = (unsigned long long)s_src[r * M_SSTR + 2 * gq] | ((unsigned long long)s_src[r * M_SSTR + 2 * gq + 1] << 32);
                unsigned long long t;
                t = (x ^ (x >> 7)) & 0x00AA00AA00AA00AAull; x = x ^ t ^ (t << 7);
                t = (x ^ (x >> 14)) & 0x0000CCCC0000CCCCull; x = x ^ t ^ (t << 14);
                t = (x ^ (x >> 28)) & 0x00000000F0F0F0F0ull; x = x ^ t ^ (t << 28);
#pragma unroll
                for (int p = 0; p < 8; p++) plb[((size_t)p * M_ROWS + r) * 8 + gq] = (uint8_t)(x >> (8 * p));
                dacc |= x ^ (x >> 8);
            }
            unsigned diff = 0;
#pragma unroll
            for (int p = 0; p < 7; p++) diff |= ((dacc >> (8 * p)) & 0xffull) ? (1u << p) : 0u;
            if (diff) atomicOr(&s_differs, diff);
        }
        __syncthreads();
        // planes that get a round: plane 7 and every plane that differs from its upper neighbour (block-uniform)
        const unsigned live = (unsigned)__builtin_amdgcn_readfirstlane((int)(s_differs | 0x80u));
        unsigned live_list = 0u;
        int nlive = 0;
        for (int p = 7; p >= 0; p--) if ((live >> p) & 1u) { live_list |= (unsigned)p << (4 * nlive); nlive++; }
        // thread -> (half, row): the two halves of a row are 128 threads apart, so a wavefront is one half of 64 consecutive rows
        const int half = tid >> 7, row = tid & (MT_H - 1);
        const int xo = x0 + 24 * half, y = y0 + row;
        if (y < h && xo < w) {
            unsigned res[8], o[6];
            uint8_t* p5 = out5 + (size_t)b * g.slot + rowoff(y, g.pitch) + xo;
            uint8_t* p7 = out7 + (size_t)b * g.slot + rowoff(y, g.pitch) + xo;
            bs_median_row<7>(s_pl, row, 24 * half, nlive, live_list, s_src);
            bs_collect(s_src, live, 1, res);                               // window bit 4 (the first output) sits at bit 4 - 3
            bs_planes_to_bytes(res, o);
            bs_store24(p7, o, xo, w);
            bs_median_row<5>(s_pl, row, 24 * half, nlive, live_list, s_src);
            bs_collect(s_src, live, 2, res);
            bs_planes_to_bytes(res, o);
            bs_store24(p5, o, xo, w);
        }
        if (kn < 0) break;
        k = kn;
        __syncthreads();                               // the results (in s_src) have been read: the next tile may be stored
    }
}

}  // namespace i2s
