// Canny (img2sgf.py:162 and the Canny inside every cv.HoughCircles call, :180) as integer HIP kernels.
// Follows OpenCV canny.cpp: Sobel 3x3 (CV_16S, BORDER_REPLICATE), L1 magnitude, per-pixel max-magnitude
// channel for colour input, non-maximum suppression with the TG22 fixed-point sectors, strict '>'
// thresholds, 8-connected hysteresis.  Map values: 0 = weak candidate, 1 = no edge, 2 = edge.
#pragma once
#include "i2s_types.h"
#include "tile_io.h"

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

// Sobel + NMS on single-channel PLANES, 4 pixels per thread with dword LDS traffic.
//   main_mode == 0: HoughCircles' internal Canny of variants [v_first, v_first + gridDim.z / nb): plane v -> map 1 + v.
//   main_mode == 1: the main Canny (img2sgf.py:162) of greyscale sources: plane 0 (grey == source) -> map 0
//                   (colour sources go through k_sobel_nms_src<3>).
// grid (tiles_x, tiles_y, nb * nvariants), block 256, tile 64 x 32 outputs.
// planes = variant plane 0 base, maps = map 0 base.
__global__ __launch_bounds__(256) void k_sobel_nms_planes(const ImgDesc* __restrict__ desc, Geo g, const uint8_t* __restrict__ planes,
                                                          uint8_t* __restrict__ maps, int v_first, int low, int high, int main_mode)
{
    constexpr int SROWS = CT_H + 4, SWORDS = CT_W / 4 + 4, SSTR = SWORDS + 1;   // source rows y0-2.., x0-8 .. x0+72
    constexpr int MROWS = CT_H + 2, MSTRIPS = CT_W / 4 + 2, MSTR = 2 * MSTRIPS + 1;   // mag rows y0-1.., x0-4 .. x0+68 (u16 pairs)
    __shared__ unsigned s_src[SROWS * SSTR];
    __shared__ unsigned s_mag[MROWS * MSTR];
    const int b = blockIdx.z % g.nb;
    const int v = main_mode ? 0 : v_first + blockIdx.z / g.nb;
    if (main_mode && desc[b].cn != 1) return;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    const uint8_t* plane = planes + ((size_t)v * g.nb + b) * g.slot;
    uint8_t* mp = maps + ((size_t)(main_mode ? 0 : 1 + v) * g.nb + b) * g.slot;
    load_tile_words<SROWS, SWORDS, SSTR, 256, BORDER_REPL>(s_src, plane, g.pitch, w, h, x0 - 8, y0 - 2, tid);
    __syncthreads();
    // gradient strips: strip (ry, s) covers pixels x = x0 - 4 + 4s .. +3 of image row y0 - 1 + ry
    constexpr int NSTRIPS = MROWS * MSTRIPS;          // 34 * 18 = 612
    constexpr int PER = (NSTRIPS + 255) / 256;        // 3
    short gdx[PER][4], gdy[PER][4];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = tid + k * 256;
        if (i < NSTRIPS) {
            const int ry = i / MSTRIPS, s = i - ry * MSTRIPS;
            int r0[6], r1[6], r2[6];
            const unsigned* p0 = s_src + ry * SSTR + s;
            unpack6(p0[0], p0[1], p0[2], r0);
            unpack6(p0[SSTR], p0[SSTR + 1], p0[SSTR + 2], r1);
            unpack6(p0[2 * SSTR], p0[2 * SSTR + 1], p0[2 * SSTR + 2], r2);
            int col[6], dif[6];
#pragma unroll
            for (int j = 0; j < 6; j++) { col[j] = r0[j] + 2 * r1[j] + r2[j]; dif[j] = r2[j] - r0[j]; }
            const int gy = y0 - 1 + ry, gx0 = x0 - 4 + 4 * s;
            unsigned m[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int dx = col[q + 2] - col[q], dy = dif[q] + 2 * dif[q + 1] + dif[q + 2];
                const int gx = gx0 + q;
                if (gx < 0 || gx >= w || gy < 0 || gy >= h) { dx = 0; dy = 0; }
                gdx[k][q] = (short)dx; gdy[k][q] = (short)dy;
                m[q] = (unsigned)(iabs_(dx) + iabs_(dy));
            }
            s_mag[ry * MSTR + 2 * s] = m[0] | (m[1] << 16);
            s_mag[ry * MSTR + 2 * s + 1] = m[2] | (m[3] << 16);
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = tid + k * 256;
        if (i >= NSTRIPS) continue;
        const int ry = i / MSTRIPS, s = i - ry * MSTRIPS;
        if (ry < 1 || ry > CT_H || s < 1 || s > CT_W / 4) continue;      // apron strips only feed neighbours
        const int gy = y0 - 1 + ry, gx0 = x0 - 4 + 4 * s;
        if (gy >= h || gx0 >= w) continue;
        // magnitudes of rows ry-1, ry, ry+1 at columns -1 .. 4 of the strip
        int mg[3][6];
#pragma unroll
        for (int rr = 0; rr < 3; rr++) {
            const unsigned* pm = s_mag + (ry - 1 + rr) * MSTR + 2 * s;
            const unsigned a = pm[-1], b0 = pm[0], b1 = pm[1], c = pm[2];
            mg[rr][0] = (int)(a >> 16); mg[rr][1] = (int)(b0 & 0xffffu); mg[rr][2] = (int)(b0 >> 16);
            mg[rr][3] = (int)(b1 & 0xffffu); mg[rr][4] = (int)(b1 >> 16); mg[rr][5] = (int)(c & 0xffffu);
        }
        unsigned outw = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int mcur = mg[1][q + 1];
            unsigned o = 1;
            if (mcur > low) {
                const int xs = gdx[k][q], ys = gdy[k][q];
                const int ax = iabs_(xs), ay = iabs_(ys) << 15;
                const int tg22x = ax * 13573;
                bool keep;
                if (ay < tg22x) keep = mcur > mg[1][q] && mcur >= mg[1][q + 2];
                else {
                    const int tg67x = tg22x + (ax << 16);
                    if (ay > tg67x) keep = mcur > mg[0][q + 1] && mcur >= mg[2][q + 1];
                    else {
                        const bool neg = (xs ^ ys) < 0;       // s = neg ? -1 : 1 -> compare (row-1, x-s) and (row+1, x+s)
                        keep = neg ? (mcur > mg[0][q + 2] && mcur > mg[2][q]) : (mcur > mg[0][q] && mcur > mg[2][q + 2]);
                    }
                }
                if (keep) o = (mcur > high) ? 2u : 0u;
            }
            outw |= o << (8 * q);
        }
        uint8_t* dstp = mp + (size_t)gy * g.pitch + gx0;
        if (gx0 + 3 < w) *reinterpret_cast<unsigned*>(dstp) = outw;
        else for (int q = 0; q < 4 && gx0 + q < w; q++) dstp[q] = (uint8_t)(outw >> (8 * q));
    }
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
