// Canny (img2sgf.py:162 and the Canny inside every cv.HoughCircles call, :180) as integer HIP kernels.
// Follows OpenCV canny.cpp: Sobel 3x3 (CV_16S, BORDER_REPLICATE), L1 magnitude, per-pixel max-magnitude
// channel for colour input, non-maximum suppression with the TG22 fixed-point sectors, strict '>'
// thresholds, 8-connected hysteresis.  Map values: 0 = weak candidate, 1 = no edge, 2 = edge.
#pragma once
#include "i2s_types.h"

namespace i2s {

constexpr int CT_W = 64;   // NMS output tile
constexpr int CT_H = 32;
constexpr int HT = 64;     // hysteresis tile (square)

// Sobel + NMS of one tile.  src(y,x,c) = sp[y*sstride + x*CN + c].
template <int CN>
__device__ __forceinline__ void sobel_nms_tile(const uint8_t* __restrict__ sp, int sstride, int w, int h,
                                               int low, int high, uint8_t* __restrict__ mp, int mpitch)
{
    constexpr int SW = CT_W + 4, SH = CT_H + 4;   // source tile with 2-px apron
    constexpr int MW = CT_W + 2, MH = CT_H + 2;   // gradient tile with 1-px apron
    __shared__ uint8_t s_src[SH][SW * CN + 4];
    __shared__ short s_dx[MH][MW], s_dy[MH][MW];
    __shared__ unsigned short s_mag[MH][MW];
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    for (int i = tid; i < SH * SW; i += 256) {
        const int ly = i / SW, lx = i - ly * SW;
        const int gy = iclamp(y0 + ly - 2, 0, h - 1), gx = iclamp(x0 + lx - 2, 0, w - 1);
        const uint8_t* p = sp + (size_t)gy * sstride + (size_t)gx * CN;
#pragma unroll
        for (int c = 0; c < CN; c++) s_src[ly][lx * CN + c] = p[c];
    }
    __syncthreads();
    for (int i = tid; i < MH * MW; i += 256) {
        const int ly = i / MW, lx = i - ly * MW;
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        int bdx = 0, bdy = 0, bm = 0;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) {
#pragma unroll
            for (int c = 0; c < CN; c++) {
                // centre of the 3x3 window in s_src is (ly+1, lx+1)
                const int a = s_src[ly][lx * CN + c], bb = s_src[ly][(lx + 1) * CN + c], cc = s_src[ly][(lx + 2) * CN + c];
                const int d = s_src[ly + 1][lx * CN + c], f = s_src[ly + 1][(lx + 2) * CN + c];
                const int gg = s_src[ly + 2][lx * CN + c], hh = s_src[ly + 2][(lx + 1) * CN + c], ii = s_src[ly + 2][(lx + 2) * CN + c];
                const int dx = (cc + 2 * f + ii) - (a + 2 * d + gg);
                const int dy = (gg + 2 * hh + ii) - (a + 2 * bb + cc);
                const int m = iabs_(dx) + iabs_(dy);
                if (c == 0 || m > bm) { bdx = dx; bdy = dy; bm = m; }
            }
        }
        s_dx[ly][lx] = (short)bdx; s_dy[ly][lx] = (short)bdy; s_mag[ly][lx] = (unsigned short)bm;
    }
    __syncthreads();
    for (int i = tid; i < CT_H * CT_W; i += 256) {
        const int ly = i / CT_W, lx = i - ly * CT_W;
        const int gx = x0 + lx, gy = y0 + ly;
        if (gx >= w || gy >= h) continue;
        const int cy = ly + 1, cx = lx + 1;
        const int m = s_mag[cy][cx];
        uint8_t out = 1;
        if (m > low) {
            const int xs = s_dx[cy][cx], ys = s_dy[cy][cx];
            const int ax = iabs_(xs), ay = iabs_(ys) << 15;
            const int tg22x = ax * 13573;
            bool keep;
            if (ay < tg22x) {
                keep = m > s_mag[cy][cx - 1] && m >= s_mag[cy][cx + 1];
            } else {
                const int tg67x = tg22x + (ax << 16);
                if (ay > tg67x) {
                    keep = m > s_mag[cy - 1][cx] && m >= s_mag[cy + 1][cx];
                } else {
                    const int s = ((xs ^ ys) < 0) ? -1 : 1;
                    keep = m > s_mag[cy - 1][cx - s] && m > s_mag[cy + 1][cx + s];
                }
            }
            if (keep) out = (m > high) ? 2 : 0;
        }
        mp[(size_t)gy * mpitch + gx] = out;
    }
}

// Main Canny (map 0) on the source image: grid (tiles_x, tiles_y, nb).
template <int CN>
__global__ __launch_bounds__(256) void k_sobel_nms_src(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ map0,
                                                       int low, int high)
{
    const int b = blockIdx.z;
    const ImgDesc im = desc[b];
    if (im.cn != CN) return;
    sobel_nms_tile<CN>(im.src, im.sstride, im.w, im.h, low, high, map0 + (size_t)b * g.slot, g.pitch);
}

// HoughCircles' internal Canny for variants [v_first, v_first + gridDim.z / nb): grid.z = nb * nvariants.
// planes = variant planes base (variant v of image b at (v * nb + b) * slot), maps likewise for map 1+v.
__global__ __launch_bounds__(256) void k_sobel_nms_var(const ImgDesc* __restrict__ desc, Geo g, const uint8_t* __restrict__ planes,
                                                       uint8_t* __restrict__ maps, int v_first, int low, int high)
{
    const int b = blockIdx.z % g.nb;
    const int v = v_first + blockIdx.z / g.nb;
    const size_t off = ((size_t)v * g.nb + b) * g.slot;
    sobel_nms_tile<1>(planes + off, g.pitch, desc[b].w, desc[b].h, low, high, maps + off, g.pitch);
}

// One hysteresis pass over maps [m_first, m_first + gridDim.z / nb).  Each block iterates its 64x64 tile
// (with a read-only 1-px apron) to a local fixed point in LDS; the host launches passes back to back and
// every pass returns immediately once the previous pass changed nothing (flags[pass-1] == 0).
// maps points at map 0; map m of image b at (m * nb + b) * slot.
__global__ __launch_bounds__(256) void k_hysteresis(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ maps,
                                                    int m_first, int* __restrict__ flags, int pass)
{
    __shared__ uint8_t s_map[HT + 2][HT + 4];
    __shared__ int s_flag[2];
    if (pass > 0 && flags[pass - 1] == 0) return;
    const int b = blockIdx.z % g.nb;
    const int m = m_first + blockIdx.z / g.nb;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = blockIdx.x * HT, y0 = blockIdx.y * HT;
    if (x0 >= w || y0 >= h) return;
    uint8_t* mp = maps + ((size_t)m * g.nb + b) * g.slot;
    const int tid = threadIdx.x;
    if (tid < 2) s_flag[tid] = 0;
    __syncthreads();
    int weak = 0;
    for (int i = tid; i < (HT + 2) * (HT + 2); i += 256) {
        const int ly = i / (HT + 2), lx = i - ly * (HT + 2);
        const int gy = y0 + ly - 1, gx = x0 + lx - 1;
        uint8_t v = 1;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) v = mp[(size_t)gy * g.pitch + gx];
        s_map[ly][lx] = v;
        if (v == 0 && ly >= 1 && ly <= HT && lx >= 1 && lx <= HT) weak = 1;
    }
    if (weak) s_flag[0] = 1;
    __syncthreads();
    if (s_flag[0] == 0) return;   // no weak pixel in this tile: nothing can change
    __syncthreads();
    // thread owns the 4x4 patch at (py, px)
    const int py = 1 + (tid / 16) * 4, px = 1 + (tid % 16) * 4;
    bool any_change = false;
    for (int iter = 0; iter < HT * HT; iter++) {
        if (tid == 0) s_flag[iter & 1] = 0;
        __syncthreads();
        bool changed = false;
        for (int dy = 0; dy < 4; dy++)
            for (int dx = 0; dx < 4; dx++) {
                const int y = py + dy, x = px + dx;
                if (s_map[y][x] != 0) continue;
                if (s_map[y - 1][x - 1] == 2 || s_map[y - 1][x] == 2 || s_map[y - 1][x + 1] == 2 ||
                    s_map[y][x - 1] == 2 || s_map[y][x + 1] == 2 ||
                    s_map[y + 1][x - 1] == 2 || s_map[y + 1][x] == 2 || s_map[y + 1][x + 1] == 2) {
                    s_map[y][x] = 2;
                    changed = true;
                }
            }
        if (changed) { s_flag[iter & 1] = 1; any_change = true; }
        __syncthreads();
        if (s_flag[iter & 1] == 0) break;
    }
    if (any_change) {
        for (int dy = 0; dy < 4; dy++)
            for (int dx = 0; dx < 4; dx++) {
                const int y = py + dy, x = px + dx;
                const int gy = y0 + y - 1, gx = x0 + x - 1;
                if (gx < w && gy < h && s_map[y][x] == 2) mp[(size_t)gy * g.pitch + gx] = 2;
            }
        flags[pass] = 1;
    }
}

// edges = 255 where map0 == 2 else 0 (img2sgf.py:162 output; also variant plane 1 and the erase target).
__global__ __launch_bounds__(256) void k_edges_from_map(const ImgDesc* __restrict__ desc, Geo g,
                                                        const uint8_t* __restrict__ map0, uint8_t* __restrict__ edges)
{
    const int b = blockIdx.z;
    const int w = desc[b].w, h = desc[b].h;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int x0 = (blockIdx.x * 64 + threadIdx.x) * 4;
    if (y >= h || x0 >= w) return;
    const uint8_t* mp = map0 + (size_t)b * g.slot + (size_t)y * g.pitch;
    uint8_t* e = edges + (size_t)b * g.slot + (size_t)y * g.pitch;
    for (int i = 0; i < 4; i++) {
        const int x = x0 + i;
        if (x < w) e[x] = mp[x] == 2 ? 255 : 0;
    }
}

}  // namespace i2s
