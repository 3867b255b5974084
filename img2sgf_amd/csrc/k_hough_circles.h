// cv.HoughCircles(img, HOUGH_GRADIENT, dp=1, minDist, param1, param2, minRadius, maxRadius) (img2sgf.py:180)
// after OpenCV hough.cpp HoughCirclesGradient (>= 3.4.2 / 4.x), restructured for CDNA4:
//   k_vote_centres  : the 2-D accumulator never exists in HBM.  Each workgroup owns a 126x126 block of
//                     accumulator cells (+1-cell apron) as a 64 KB LDS tile, gathers the edge pixels whose
//                     gradient rays can reach it (LDS compaction list), casts the votes with LDS atomics and
//                     tests the 4-neighbour local-maximum rule in place; only centre candidates leave the CU.
//   k_radius        : one wavefront per centre: 10-bins-per-pixel radius histogram of the edge bitmap in LDS.
//   k_circles_final : per (image, variant) bitonic sort by OpenCV's total order + greedy min-dist pass.
#pragma once
#include "i2s_types.h"

namespace i2s {

constexpr int VT = 126;          // accumulator cells per tile side (interior)
constexpr int VL = VT + 2;       // LDS tile side incl. apron
constexpr int VSTRIP = 16;       // candidate rows gathered per compaction round
constexpr int VLIST_CAP = VSTRIP * (VL + 2 * 30);   // worst case: every scanned pixel is an edge

// Sobel 3x3 with BORDER_REPLICATE at one pixel of a single-channel plane.
__device__ __forceinline__ void sobel_at(const uint8_t* __restrict__ p, int pitch, int w, int h, int x, int y, int& dx, int& dy)
{
    const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
    const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
    const uint8_t* r0 = p + (size_t)ym * pitch;
    const uint8_t* r1 = p + (size_t)y * pitch;
    const uint8_t* r2 = p + (size_t)yp * pitch;
    const int a = r0[xm], b = r0[x], c = r0[xp], d = r1[xm], f = r1[xp], gg = r2[xm], hh = r2[x], ii = r2[xp];
    dx = (c + 2 * f + ii) - (a + 2 * d + gg);
    dy = (gg + 2 * hh + ii) - (a + 2 * b + c);
}

// grid (tiles_x, tiles_y, nb * NVAR).  planes/maps: variant v of image b at (v * nb + b) * slot (maps = map 1 base).
// cent_list[(b * NVAR + v) * CENT_CAP + i] = x | y << 16 of an accumulator local maximum; cent_count likewise.
// dbg_acc (optional): dense int32 accumulator, cell (x,y) of (b,v) at ((b * NVAR + v) * hmax + y) * pitch + x.
__global__ __launch_bounds__(256) void k_vote_centres(const ImgDesc* __restrict__ desc, Geo g,
                                                      const uint8_t* __restrict__ planes, const uint8_t* __restrict__ maps,
                                                      int min_r, int max_r, int acc_thr,
                                                      unsigned* __restrict__ cent_list, int* __restrict__ cent_count,
                                                      int* __restrict__ dbg_acc)
{
    __shared__ unsigned s_acc[VL * VL];
    __shared__ unsigned s_list[VLIST_CAP];
    __shared__ int s_n;
    const int b = blockIdx.z / NVAR, v = blockIdx.z % NVAR;
    const int w = desc[b].w, h = desc[b].h;
    const int cx0 = blockIdx.x * VT, cy0 = blockIdx.y * VT;    // first interior cell
    if (cx0 >= w || cy0 >= h) return;
    const int tid = threadIdx.x;
    const size_t off = ((size_t)v * g.nb + b) * g.slot;
    const uint8_t* plane = planes + off;
    const uint8_t* map = maps + off;
    for (int i = tid; i < VL * VL; i += 256) s_acc[i] = 0;
    // LDS tile covers cells [lx0, lx0 + VL) x [ly0, ly0 + VL)
    const int lx0 = cx0 - 1, ly0 = cy0 - 1;
    // edge pixels that can vote into the tile: within max_r of it (|step| <= 1024 per radius unit)
    const int rx0 = imax(lx0 - max_r, 0), rx1 = imin(lx0 + VL + max_r, w);
    const int ry0 = imax(ly0 - max_r, 0), ry1 = imin(ly0 + VL + max_r, h);
    const int cw = rx1 - rx0;
    for (int sy = ry0; sy < ry1; sy += VSTRIP) {
        if (tid == 0) s_n = 0;
        __syncthreads();
        const int rows = imin(VSTRIP, ry1 - sy);
        for (int i = tid; i < rows * cw; i += 256) {
            const int ly = i / cw, lx = i - ly * cw;
            const int x = rx0 + lx, y = sy + ly;
            if (map[(size_t)y * g.pitch + x] == 2) {
                const int k = atomicAdd(&s_n, 1);
                s_list[k] = (unsigned)x | ((unsigned)y << 16);
            }
        }
        __syncthreads();
        const int n = s_n;
        for (int k = tid; k < n; k += 256) {
            const unsigned e = s_list[k];
            const int x = (int)(e & 0xffffu), y = (int)(e >> 16);
            int dx, dy;
            sobel_at(plane, g.pitch, w, h, x, y, dx, dy);
            if (dx == 0 && dy == 0) continue;
            const float vx = (float)dx, vy = (float)dy;
            const float mag = sqrtf(vx * vx + vy * vy);
            if (mag < 1.0f) continue;
            int sx = __float2int_rn((vx * 1.0f) * 1024.0f / mag);
            int sy2 = __float2int_rn((vy * 1.0f) * 1024.0f / mag);
            const int xb = x * 1024, yb = y * 1024;
            for (int k1 = 0; k1 < 2; k1++) {
                int x1 = xb + min_r * sx, y1 = yb + min_r * sy2;
                for (int r = min_r; r <= max_r; x1 += sx, y1 += sy2, r++) {
                    const int x2 = x1 >> 10, y2 = y1 >> 10;
                    if ((unsigned)x2 >= (unsigned)w || (unsigned)y2 >= (unsigned)h) break;
                    const unsigned tx = (unsigned)(x2 - lx0), ty = (unsigned)(y2 - ly0);
                    if (tx < (unsigned)VL && ty < (unsigned)VL) atomicAdd(&s_acc[ty * VL + tx], 1u);
                }
                sx = -sx; sy2 = -sy2;
            }
        }
        __syncthreads();
    }
    // centre candidates: cells (x,y), 1 <= x <= w-1, 1 <= y <= h-1 (OpenCV scans padded rows 1..H, cols 1..W
    // of an accumulator whose votes sit at unpadded indices; cells x == W or y == H hold no votes).
    const int bv = b * NVAR + v;
    for (int i = tid; i < VT * VT; i += 256) {
        const int ty = i / VT + 1, tx = i - (ty - 1) * VT + 1;
        const int x = lx0 + tx, y = ly0 + ty;
        if (x >= w || y >= h) continue;
        const unsigned a = s_acc[ty * VL + tx];
        if (dbg_acc) dbg_acc[((size_t)bv * g.hmax + y) * g.pitch + x] = (int)a;
        if (x < 1 || y < 1) continue;
        if ((int)a > acc_thr && a > s_acc[ty * VL + tx - 1] && a >= s_acc[ty * VL + tx + 1] &&
            a > s_acc[(ty - 1) * VL + tx] && a >= s_acc[(ty + 1) * VL + tx]) {
            const int k = atomicAdd(&cent_count[bv], 1);
            if (k < CENT_CAP) cent_list[(size_t)bv * CENT_CAP + k] = (unsigned)x | ((unsigned)y << 16);
        }
    }
}

// Sort key of an estimated circle; ascending key order == OpenCV's cmpAccum order
// (accum desc, radius desc, x asc, y asc).  s = upbin + j of the radius histogram scan (radius = s/20 + min_r).
__device__ __forceinline__ unsigned long long est_key(int acc, int s, int x, int y)
{
    return ((unsigned long long)(4095 - acc) << 42) | ((unsigned long long)(1023 - s) << 32) |
           ((unsigned long long)(unsigned)x << 16) | (unsigned long long)(unsigned)y;
}

constexpr int RAD_BINS_MAX = 320;

// grid (RAD_GX, nb * NVAR), block 256 = 4 wavefronts, one centre per wavefront per round.
// est_keys[(b*NVAR+v) * EST_CAP + i], est_count[b*NVAR+v].
__global__ __launch_bounds__(256) void k_radius(const ImgDesc* __restrict__ desc, Geo g, const uint8_t* __restrict__ maps,
                                                const unsigned* __restrict__ cent_list, const int* __restrict__ cent_count,
                                                int min_r, int max_r, int acc_thr,
                                                unsigned long long* __restrict__ est_keys, int* __restrict__ est_count)
{
    __shared__ int s_bins[4][RAD_BINS_MAX];
    const int bv = blockIdx.y;
    const int b = bv / NVAR, v = bv % NVAR;
    const int w = desc[b].w, h = desc[b].h;
    const int n = imin(cent_count[bv], CENT_CAP);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint8_t* map = maps + ((size_t)v * g.nb + b) * g.slot;
    const int nBinsPerDr = 10;
    int nBins = __float2int_rn((float)(max_r - min_r) / 1.0f * (float)nBinsPerDr);
    if (nBins < 1) nBins = 1;
    const float minR2 = (float)min_r * (float)min_r, maxR2 = (float)max_r * (float)max_r;
    for (int c0 = blockIdx.x * 4; c0 < n; c0 += gridDim.x * 4) {
        const int c = c0 + wave;
        const bool live = c < n;
        for (int i = lane; i < nBins; i += 64) s_bins[wave][i] = 0;
        __syncthreads();
        int cxi = 0, cyi = 0;
        if (live) {
            const unsigned e = cent_list[(size_t)bv * CENT_CAP + c];
            cxi = (int)(e & 0xffffu); cyi = (int)(e >> 16);
            const float cx = ((float)cxi + 0.5f) * 1.0f, cy = ((float)cyi + 0.5f) * 1.0f;
            // NZPointSet::filterCircles box: [int(c - (maxR+1)), int(c + (maxR+1))) clipped to the image
            const int rOuter = max_r + 1;
            const int bx0 = imax((int)(cx - (float)rOuter), 0), bx1 = imin((int)(cx + (float)rOuter), w);
            const int by0 = imax((int)(cy - (float)rOuter), 0), by1 = imin((int)(cy + (float)rOuter), h);
            const int bw = bx1 - bx0, npx = bw * (by1 - by0);
            for (int i = lane; i < npx; i += 64) {
                const int yy = i / bw, xx = i - yy * bw;
                const int px = bx0 + xx, py = by0 + yy;
                if (map[(size_t)py * g.pitch + px] != 2) continue;
                const float ddx = cx - (float)px, ddy = cy - (float)py;
                const float r2 = ddx * ddx + ddy * ddy;
                if (minR2 <= r2 && r2 <= maxR2) {
                    const float d = sqrtf(r2);
                    int bin = __float2int_rn((d - (float)min_r) / 1.0f * (float)nBinsPerDr);
                    bin = imax(0, imin(nBins - 1, bin));
                    atomicAdd(&s_bins[wave][bin], 1);
                }
            }
        }
        __syncthreads();
        if (live && lane == 0) {
            const int* bins = s_bins[wave];
            int maxCount = 0, sBest = 0;
            float rBest = 0.f;
            for (int j = nBins - 1; j > 0; j--) {
                if (bins[j]) {
                    const int upbin = j;
                    int curCount = 0;
                    for (; j > upbin - nBinsPerDr && j >= 0; j--) curCount += bins[j];
                    const float rCur = (float)(upbin + j) / 2.f / (float)nBinsPerDr * 1.0f + (float)min_r;
                    if (((float)curCount * rBest >= (float)maxCount * rCur) || (rBest < 1.1920929e-07f && curCount >= maxCount)) {
                        rBest = rCur; maxCount = curCount; sBest = upbin + j;
                    }
                }
            }
            if (maxCount > acc_thr) {
                const int k = atomicAdd(&est_count[bv], 1);
                if (k < EST_CAP) est_keys[(size_t)bv * EST_CAP + k] = est_key(imin(maxCount, 4095), sBest, cxi, cyi);
            }
        }
        __syncthreads();
    }
}

// grid (nb * NVAR), block 256.  Sorts the estimates, runs RemoveOverlaps (greedy: keep a circle iff it is at
// least min_dist from every circle already kept) and writes circles (x, y, r) in output order.
// vcirc[(bv * VCIRC_CAP + i) * 3], vcount[bv]; overflow[b] is set when a capacity was exceeded.
__global__ __launch_bounds__(256) void k_circles_final(Geo g, const unsigned long long* __restrict__ est_keys,
                                                       const int* __restrict__ est_count, const int* __restrict__ cent_count,
                                                       float min_dist, int min_r,
                                                       float* __restrict__ vcirc, int* __restrict__ vcount, int* __restrict__ overflow)
{
    __shared__ unsigned long long s_key[EST_CAP];
    __shared__ short s_kx[VCIRC_CAP], s_ky[VCIRC_CAP];
    const int bv = blockIdx.x;
    const int b = bv / NVAR;
    const int tid = threadIdx.x;
    int n = est_count[bv];
    if (cent_count[bv] > CENT_CAP || n > EST_CAP) {
        if (tid == 0) { overflow[b] = 1; vcount[bv] = 0; }
        return;
    }
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += 256) s_key[i] = i < n ? est_keys[(size_t)bv * EST_CAP + i] : ~0ull;
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long a = s_key[i], c = s_key[ixj];
                    const bool up = (i & k) == 0;
                    if ((a > c) == up) { s_key[i] = c; s_key[ixj] = a; }
                }
            }
            __syncthreads();
        }
    // greedy min-dist pass by wavefront 0 (sequential over candidates, 64 kept circles per step)
    if (tid < 64) {
        const float md2 = min_dist * min_dist;
        int kept = 0;
        bool over = false;
        for (int i = 0; i < n; i++) {
            const unsigned long long key = s_key[i];
            const int x = (int)((key >> 16) & 0xffffu), y = (int)(key & 0xffffu);
            bool bad = false;
            for (int j0 = 0; j0 < kept; j0 += 64) {
                const int j = j0 + tid;
                bool hit = false;
                if (j < kept) {
                    const float ddx = (float)(x - s_kx[j]), ddy = (float)(y - s_ky[j]);
                    hit = ddx * ddx + ddy * ddy < md2;
                }
                if (__ballot(hit) != 0ull) { bad = true; break; }
            }
            if (!bad) {
                if (kept >= VCIRC_CAP) { over = true; break; }
                if (tid == 0) {
                    s_kx[kept] = (short)x; s_ky[kept] = (short)y;
                    const int s = 1023 - (int)((key >> 32) & 0x3ffu);
                    float* o = vcirc + ((size_t)bv * VCIRC_CAP + kept) * 3;
                    o[0] = ((float)x + 0.5f) * 1.0f;
                    o[1] = ((float)y + 0.5f) * 1.0f;
                    o[2] = (float)s / 2.f / 10.f * 1.0f + (float)min_r;
                }
                kept++;
            }
        }
        if (tid == 0) { vcount[bv] = over ? 0 : kept; if (over) overflow[b] = 1; }
    }
}

}  // namespace i2s
