// PROTOTYPE for counting -- not part of the product, never loaded by it.
// Exact 5x5 and 7x7 medians (cv.medianBlur, img2sgf.py:174: BORDER_REPLICATE) as separable sorting networks with shared columns:
// the formulation DESIGN section 4's floor argument for k_median57 (bit-sliced radix selection) did not cover.
//
//   * two pixels per register: the low half holds a pixel of column strip A, the high half the pixel 248 columns to its right
//     (strip B), so that the two halves have identical geometry and v_pk_min_u16 / v_pk_max_u16 work on two outputs at once;
//   * a lane owns 4 neighbouring columns; a 64-lane workgroup covers 2 x 248 output columns (lanes 0 and 63 are apron lanes);
//   * per iteration 2 output rows from 8 input rows; the min / max network itself is generated (gen.py -> median_net_body.inc) from
//     tools/median_network.py, which checks every network it builds against numpy;
//   * values that depend on ONE neighbouring lane's columns only (sorted columns, merged column pairs) come from that lane through LDS.
//
// Build for counting:   hipcc --offload-arch=gfx950 -O3 -S -o median_net.s tools/experiments/median_net/median_net.hip   (count.py)
// Build for checking:   g++ -I tests/emu ... (check.py runs it on the fiber emulation against numpy's medians)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "median_net_counts.h"

#define XCH_STRIDE 66
#define STRIP 248

#ifdef HIPEMU
static inline unsigned pkmin(unsigned a, unsigned b)
{
    const unsigned l = (a & 0xffff) < (b & 0xffff) ? (a & 0xffff) : (b & 0xffff), h = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
    return l | h << 16;
}
static inline unsigned pkmax(unsigned a, unsigned b)
{
    const unsigned l = (a & 0xffff) > (b & 0xffff) ? (a & 0xffff) : (b & 0xffff), h = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
    return l | h << 16;
}
#else
typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ unsigned pkmin(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
static __device__ __forceinline__ unsigned pkmax(unsigned a, unsigned b)
{
    unsigned r;
    asm("v_pk_max_u16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#endif

// four pixels of row y starting at column x (multiple of 4), BORDER_REPLICATE
static __device__ __forceinline__ unsigned load4(const uint8_t* src, int pitch, int w, int h, int x, int y)
{
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    const uint8_t* row = src + (size_t)y * pitch;
    if (x >= 0 && x + 3 < w) return *(const unsigned*)(row + x);
    unsigned v = 0;
    for (int i = 0; i < 4; i++) {
        int xx = x + i;
        xx = xx < 0 ? 0 : (xx >= w ? w - 1 : xx);
        v |= (unsigned)row[xx] << (8 * i);
    }
    return v;
}

// grid: (ceil(w / 496), ceil(h / rows_per_block)); block: 64.  pitch and w multiples of 4.
extern "C" __global__ void __launch_bounds__(64) k_median57_net(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst5, uint8_t* __restrict__ dst7,
                                                                int pitch, int w, int h, int rows_per_block)
{
    __shared__ unsigned xch[XCH_SLOTS * XCH_STRIDE];
    const int lane = threadIdx.x;
    const int xa = blockIdx.x * (2 * STRIP) + (lane - 1) * 4, xb = xa + STRIP;        // this lane's columns in strip A / strip B
    const int y_first = blockIdx.y * rows_per_block;
    const int y_end = y_first + rows_per_block < h ? y_first + rows_per_block : h;
    unsigned p[8][4];                                    // input rows y-3 .. y+4, packed (A | B << 16) per column
    auto load_row = [&](int slot, int y) {
        const unsigned a = load4(src, pitch, w, h, xa, y), b = load4(src, pitch, w, h, xb, y);
        for (int c = 0; c < 4; c++) p[slot][c] = ((a >> (8 * c)) & 0xff) | ((b >> (8 * c)) & 0xff) << 16;
    };
    for (int i = 0; i < 6; i++) load_row(i + 2, y_first - 3 + i);
    for (int y = y_first; y < y_end; y += 2) {
        for (int i = 0; i < 6; i++) for (int c = 0; c < 4; c++) p[i][c] = p[i + 2][c];
        load_row(6, y + 3);
        load_row(7, y + 4);
        unsigned out5[2][4], out7[2][4];
#include "median_net_body.inc"
        const bool mine = lane >= 1 && lane <= 62;
        for (int r = 0; r < 2; r++) {
            if (y + r >= h || !mine) continue;
            const unsigned a5 = (out5[r][0] & 0xff) | (out5[r][1] & 0xff) << 8 | (out5[r][2] & 0xff) << 16 | (out5[r][3] & 0xff) << 24;
            const unsigned b5 = (out5[r][0] >> 16 & 0xff) | (out5[r][1] >> 16 & 0xff) << 8 | (out5[r][2] >> 16 & 0xff) << 16 | (out5[r][3] >> 16) << 24;
            const unsigned a7 = (out7[r][0] & 0xff) | (out7[r][1] & 0xff) << 8 | (out7[r][2] & 0xff) << 16 | (out7[r][3] & 0xff) << 24;
            const unsigned b7 = (out7[r][0] >> 16 & 0xff) | (out7[r][1] >> 16 & 0xff) << 8 | (out7[r][2] >> 16 & 0xff) << 16 | (out7[r][3] >> 16) << 24;
            if (xa < w) { *(unsigned*)(dst5 + (size_t)(y + r) * pitch + xa) = a5; *(unsigned*)(dst7 + (size_t)(y + r) * pitch + xa) = a7; }
            if (xb < w) { *(unsigned*)(dst5 + (size_t)(y + r) * pitch + xb) = b5; *(unsigned*)(dst7 + (size_t)(y + r) * pitch + xb) = b7; }
        }
    }
}

#ifdef HIPEMU
// check.py: reads w h and the image from stdin (binary), writes the two median planes to stdout
#include <stdio.h>
#include <vector>
int main()
{
    int wh[2];
    if (fread(wh, 4, 2, stdin) != 2) return 1;
    const int w = wh[0], h = wh[1], pitch = (w + 3) / 4 * 4;
    std::vector<uint8_t> img((size_t)pitch * h), d5((size_t)pitch * h), d7((size_t)pitch * h);
    for (int y = 0; y < h; y++) if (fread(&img[(size_t)y * pitch], 1, w, stdin) != (size_t)w) return 1;
    const int rpb = 16;
    hipLaunchKernelGGL(k_median57_net, dim3((w + 2 * STRIP - 1) / (2 * STRIP), (h + rpb - 1) / rpb), dim3(64), 0, 0, img.data(), d5.data(), d7.data(), pitch, w, h, rpb);
    for (int y = 0; y < h; y++) fwrite(&d5[(size_t)y * pitch], 1, w, stdout);
    for (int y = 0; y < h; y++) fwrite(&d7[(size_t)y * pitch], 1, w, stdout);
    return 0;
}
#endif
