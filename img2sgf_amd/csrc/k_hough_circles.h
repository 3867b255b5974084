// cv.HoughCircles(img, HOUGH_GRADIENT, dp=1, minDist, param1, param2, minRadius, maxRadius) (img2sgf.py:180)
// after OpenCV hough.cpp HoughCirclesGradient (>= 3.4.2 / 4.x), restructured for CDNA4:
//   k_edge_bins     : edge pixels -> 8-byte records (position + fixed-point unit gradient) binned by 32x32 cell.
//   k_vote_centres  : the 2-D accumulator never exists in HBM.  Each workgroup owns a 126x126 block of
//                     accumulator cells (+1-cell apron) as a 33 KB LDS tile of 16-bit counters, streams the edge
//                     bins within reach, casts the votes with LDS atomics (64 (record, direction) rays per
//                     wavefront, one lane each, stepping through the radii together) and tests the 4-neighbour
//                     local-maximum rule in place; only centre candidates leave the CU.
//   k_radius        : one wavefront per centre: 10-bins-per-pixel radius histogram of the edge bitmap in LDS.
//   k_circles_final : per (image, variant) bitonic sort by OpenCV's total order + greedy min-dist pass.
#pragma once
#include "i2s_types.h"

namespace i2s {

constexpr int VT = 126;          // accumulator cells per tile side (interior)
constexpr int VL = VT + 2;       // LDS tile side incl. apron
constexpr int VASTR = 129;       // dword row stride of the LDS tile (odd: vertical rays spread over the banks)
constexpr int VTHREADS = 512;
constexpr int EB = 32;           // edge bins: EB x EB pixel cells
#ifndef I2S_EB_CAP
#define I2S_EB_CAP (EB * EB)
#endif
constexpr int EB_CAP = I2S_EB_CAP;  // worst case: every pixel of a bin is an edge
constexpr int VRING = 192;            // per-wave item ring of k_vote_centres: < 64 waiting + <= 128 new per round
constexpr int EBB_X = 4, EBB_Y = 1;   // bins per k_edge_bins workgroup (128 x 32 pixels)
// threads, rows per load round, 16-byte loads per thread.  Measured (us per diagram): 256 threads on 128 x 64 pixels 4.47, on 128 x 32: 5.33, on
// 128 x 128: 6.2; 128 threads on 128 x 32 (the same two loads per thread, twice the workgroups in flight): 4.27; 64 threads: 5.28
constexpr int EBT = 128, EB_RPI = EBT / 8, EB_NLD = EBB_Y * EB / EB_RPI;

__device__ __forceinline__ unsigned umax_(unsigned a, unsigned b) { return a > b ? a : b; }

// Sobel 3x3 with BORDER_REPLICATE at one pixel of a single-channel plane.
__device__ __forceinline__ void sobel_at(const uint8_t* __restrict__ p, int pitch, int w, int h, int x, int y, int& dx, int& dy)
{
    const int xm = x > 0 ? x - 1 : 0, xp = x < w - 1 ? x + 1 : w - 1;
    const int ym = y > 0 ? y - 1 : 0, yp = y < h - 1 ? y + 1 : h - 1;
    const uint8_t* r0 = p + rowoff(ym, pitch);
    const uint8_t* r1 = p + rowoff(y, pitch);
    const uint8_t* r2 = p + rowoff(yp, pitch);
    const int a = r0[xm], b = r0[x], c = r0[xp], d = r1[xm], f = r1[xp], gg = r2[xm], hh = r2[x], ii = r2[xp];
    dx = (c + 2 * f + ii) - (a + 2 * d + gg);
    dy = (gg + 2 * hh + ii) - (a + 2 * b + c);
}

// ---- edge bins -------------------------------------------------------------------------------------------------------
// After hysteresis, every edge pixel of a HoughCircles input becomes one 8-byte record
//   .x = x | y << 16,  .y = (sx & 0xffff) | sy << 16
// with (sx, sy) = cvRound(d * 1024 / |d|) of its Sobel gradient d (hough.cpp HoughCirclesAccumInvoker), stored in the
// bin of its 32x32-pixel cell.  The vote kernel then streams only the bins within reach of its accumulator tile.
// grid (ceil(bins_x / EBB_X) * ceil(bins_y / EBB_Y) * nb * NVAR), block EBT (32 pixels per thread).
// bin_cnt[(bv * g.bins) + by * g.bw + bx], bin_ent[... * EB_CAP + k].
__global__ __launch_bounds__(EBT) void k_edge_bins(const ImgDesc* __restrict__ desc, Geo g,
                                                   const uint8_t* __restrict__ planes, const uint8_t* __restrict__ maps,
                                                   uint2* __restrict__ bin_ent, int* __restrict__ bin_cnt, int gx, int gy)
{
    // one block = 4 x 1 bins (128 x 32 pixels); two 16-byte map loads per thread, both in flight together (the kernel is
    // latency-bound: load -> compact -> gather -> store).  Edge positions are first compacted into an LDS list (13-bit
    // tile-local coordinates) so that the gradient work (8 neighbour loads, sqrt, 2 divides) is spread evenly over the
    // block instead of serialising inside the few threads whose pixels lie on a line.
    __shared__ unsigned short s_list[EBB_X * EBB_Y * EB * EB];
    __shared__ int s_nl;
    __shared__ int s_n[EBB_X * EBB_Y];
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z / NVAR, v = tl.z % NVAR;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = tl.tx * (EBB_X * EB), y0 = tl.ty * (EBB_Y * EB);
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    const size_t off = ((size_t)v * g.nb + b) * g.slot;
    const uint8_t* plane = v == 0 ? desc[b].grey : planes + off;      // variant 0 may be the source image itself (ImgDesc::grey)
    const int ppitch = v == 0 ? desc[b].gpitch : g.pitch;
    const uint8_t* map = maps + off;
    const size_t bin0 = (size_t)(b * NVAR + v) * g.bins + (size_t)tl.ty * EBB_Y * g.bw + (size_t)tl.tx * EBB_X;
    if (tid < EBB_X * EBB_Y) s_n[tid] = 0;
    if (tid == 32) s_nl = 0;
    __syncthreads();
    {
        const int ly = tid >> 3, c16 = (tid & 7) * 16;
        const int xs = x0 + c16;
        uint4 m16[EB_NLD];
#pragma unroll
        for (int r = 0; r < EB_NLD; r++) {
            const int y = y0 + ly + r * EB_RPI;
            m16[r] = make_uint4(0u, 0u, 0u, 0u);
            if (y < h && xs < w) m16[r] = *reinterpret_cast<const uint4*>(map + rowoff(y, g.pitch) + xs);
        }
#pragma unroll
        for (int r = 0; r < EB_NLD; r++) {
            const unsigned mw[4] = {m16[r].x, m16[r].y, m16[r].z, m16[r].w};
#pragma unroll
            for (int d = 0; d < 4; d++) {
                const unsigned m4 = mw[d];
                const unsigned t = m4 ^ 0x02020202u;
                if (((t - 0x01010101u) & ~t & 0x80808080u) == 0) continue;      // no byte == 2
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int lx = c16 + 4 * d + q;
                    if (x0 + lx < w && ((m4 >> (8 * q)) & 0xffu) == 2u)
                        s_list[atomicAdd(&s_nl, 1)] = (unsigned short)(lx | ((ly + r * EB_RPI) << 7));
                }
            }
        }
    }
    __syncthreads();
    const int nl = s_nl;
    for (int i = tid; i < nl; i += EBT) {
        const int le = s_list[i];
        const int lx = le & 127, lyy = le >> 7;
        const int x = x0 + lx, y = y0 + lyy;
        int dx, dy;
        if (x >= 1 && y >= 1 && y <= h - 2 && (x <= w - 3 || (x == w - 2 && y <= h - 3))) {
            // interior pixel: the 3x3 neighbourhood as three (unaligned) dword loads instead of eight byte loads.  The
            // fourth byte of a load is pixel x + 2 (unused): for x = w - 2 it is the first byte of the next row, which
            // does not exist below the last row of a source image used in place -- that one pixel takes the byte path.
            const uint8_t* pc = plane + rowoff(y, ppitch) + (x - 1);
            unsigned r0, r1, r2;
            __builtin_memcpy(&r0, pc - ppitch, 4);
            __builtin_memcpy(&r1, pc, 4);
            __builtin_memcpy(&r2, pc + ppitch, 4);
            const int a = (int)(r0 & 0xffu), bb = (int)((r0 >> 8) & 0xffu), c = (int)((r0 >> 16) & 0xffu);
            const int d = (int)(r1 & 0xffu), f = (int)((r1 >> 16) & 0xffu);
            const int gg = (int)(r2 & 0xffu), hh = (int)((r2 >> 8) & 0xffu), ii = (int)((r2 >> 16) & 0xffu);
            dx = (c + 2 * f + ii) - (a + 2 * d + gg);
            dy = (gg + 2 * hh + ii) - (a + 2 * bb + c);
        } else {
            sobel_at(plane, ppitch, w, h, x, y, dx, dy);
        }
        if (dx == 0 && dy == 0) continue;
        const float vx = (float)dx, vy = (float)dy;
        const float mag = sqrtf(vx * vx + vy * vy);
        if (mag < 1.0f) continue;
        const int sx = __float2int_rn((vx * 1.0f) * 1024.0f / mag);
        const int sy = __float2int_rn((vy * 1.0f) * 1024.0f / mag);
        const int kbx = lx / EB, kby = lyy / EB;
        const int k = atomicAdd(&s_n[kby * EBB_X + kbx], 1);
        bin_ent[(bin0 + (size_t)kby * g.bw + kbx) * EB_CAP + k] =
            make_uint2((unsigned)x | ((unsigned)y << 16), ((unsigned)sx & 0xffffu) | ((unsigned)sy << 16));
    }
    __syncthreads();
    if (tid < EBB_X * EBB_Y) {
        const int kbx = tid % EBB_X, kby = tid / EBB_X;
        if (x0 + kbx * EB < w && y0 + kby * EB < h) bin_cnt[bin0 + (size_t)kby * g.bw + kbx] = s_n[tid];
    }
}

// Votes of up to 64 (edge record, direction) items held in a per-wave LDS ring (item = index into bin_ent | direction << 31).
// Lane l walks item l through r = min_r .. max_r: cell = ((x * 1024 + r * sx) >> 10, (y * 1024 + r * sy) >> 10), relative to
// the first valid cell of the tile, with (sx, sy) negated for the second direction.  Cells outside [0, vx_n) x [0, vy_n)
// (outside the image or the tile) are skipped, which equals OpenCV's "break at the first cell outside the image" because a
// ray leaves the convex image only once.
template <int NSTEPS>   // > 0: the number of radius steps is known at compile time (the loop is unrolled); 0: use `nsteps`
__device__ __forceinline__ void vote_walk64(const unsigned* __restrict__ ring, int count, int lane,
                                            const uint2* __restrict__ bin_ent, int vx_lo, int vy_lo,
                                            unsigned vx_n, unsigned vy_n, int offx, int offy, int min_r, int nsteps,
                                            unsigned* __restrict__ s_acc)
{
    // x, y: 22.10 fixed point relative to the LDS tile's first cell (the valid-cell origin (vx_lo, vy_lo) sits at (offx, offy) in
    // the tile), so that one unsigned compare per axis against the limits below is the whole range test and the shifted
    // coordinates index the tile directly.  In a tile on the image's left / top border this also lets through the cells of
    // column / row -1: they land in the apron, which there is only ever read as the neighbour of a column-0 / row-0 cell, and
    // those are never centre candidates (OpenCV scans rows / columns 1 ..).
    int sx = 0, sy = 0, x = -1024, y = -1024;          // idle lanes sit at cell (-1, -1): never in range
    if (lane < count) {
        const unsigned item = ring[lane];
        const uint2 e = bin_ent[item & 0x7fffffffu];   // read a moment ago by the culling pass: an L1 / L2 hit
        sx = (int)(short)(e.y & 0xffffu); sy = (int)(short)(e.y >> 16);
        if (item >> 31) { sx = -sx; sy = -sy; }
        x = (((int)(e.x & 0xffffu) - vx_lo + offx) << 10) + __mul24(min_r, sx);
        y = (((int)(e.x >> 16) - vy_lo + offy) << 10) + __mul24(min_r, sy);
    }
    const unsigned xl = (vx_n + (unsigned)offx) << 10, yl = (vy_n + (unsigned)offy) << 10;
    auto step = [&]() {
        if ((unsigned)x < xl && (unsigned)y < yl) {
            const unsigned ty = (unsigned)y >> 10;
            // the half of the dword: rows 64 .. 127 are bit 16 of y (y < 128 << 10 here); max(y & 0x10000, 1) is 0x10000 / 1 in two instructions
            atomicAdd(&s_acc[(ty & 63u) * (unsigned)VASTR + ((unsigned)x >> 10)], umax_((unsigned)y & 0x10000u, 1u));
        }
        x += sx; y += sy;
    };
    if (NSTEPS > 0) {
#pragma unroll
        for (int st = 0; st < NSTEPS; st++) step();
    } else {
        for (int st = 0; st < nsteps; st++) step();
    }
    __builtin_amdgcn_wave_barrier();
}

// grid (tiles_x, tiles_y, nb * NVAR), block 512.
// cent_list[(b * NVAR + v) * g.cent_cap + i] = x | y << 16 of an accumulator local maximum; cent_count likewise.
// dbg_acc (optional): dense int32 accumulator, cell (x,y) of (b,v) at ((b * NVAR + v) * hmax + y) * pitch + x.
//
// Votes of one edge pixel: cells ((x*1024 +- r*sx) >> 10, (y*1024 +- r*sy) >> 10), r = min_r..max_r, that lie inside the
// image (OpenCV walks r upward and breaks at the first cell outside; the walk is a straight line from inside a convex
// image, so "break" == "skip every outside cell").  The (edge, direction, r) votes are therefore independent: a wavefront
// walks 64 (edge, direction) rays at a time, one per lane (vote_walk64).  The 2-D accumulator never exists in HBM: each
// workgroup owns 126x126 cells (+1-cell apron) in LDS and tests the 4-neighbour local-maximum rule in place.
// Cell counts are 16-bit halves of LDS dwords: a cell receives at most 3 votes from each of the < 3100 edge pixels
// within max_r <= 30 of it, so a half never carries into its neighbour.  The two halves of a dword are cells 64 rows
// apart: neighbouring cells, which rays of neighbouring edge pixels hit in the same instruction, never share a dword
// (same-address LDS atomics serialise, ~4 cycles per extra lane: profiles/r01_g_lds_atomic_microbench.txt).
// (Measured: plain 32-bit cells halve the resident workgroups per CU and run 1.5x slower; a branch-free variant that
// lets out-of-tile lanes add 0 to clamped cells runs 1.4x slower because of same-address conflicts.)
template <int NSTEPS>
__global__ __launch_bounds__(512) void k_vote_centres(const ImgDesc* __restrict__ desc, Geo g,
                                                      const uint2* __restrict__ bin_ent, const int* __restrict__ bin_cnt,
                                                      int min_r, int max_r, int acc_thr,
                                                      unsigned* __restrict__ cent_list, int* __restrict__ cent_count,
                                                      int* __restrict__ dbg_acc, int gx, int gy)
{
    __shared__ unsigned s_acc[(VL / 2) * VASTR];
    __shared__ int s_ticket;
    __shared__ unsigned s_ring[VTHREADS / 64][VRING];
    __shared__ int s_fill[VTHREADS / 64];
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z / NVAR, v = tl.z % NVAR;
    const int w = desc[b].w, h = desc[b].h;
    const int cx0 = tl.tx * VT, cy0 = tl.ty * VT;    // first interior cell
    if (cx0 >= w || cy0 >= h) return;
    const int tid = threadIdx.x;
    const int bv = b * NVAR + v;
    for (int i = tid; i < (VL / 2) * VASTR; i += VTHREADS) s_acc[i] = 0;
    if (tid == 0) s_ticket = VTHREADS / 64;        // bins 0 .. 7 are taken by the waves' first round
    __syncthreads();
    // LDS tile covers cells [lx0, lx0 + VL) x [ly0, ly0 + VL); edge pixels within max_r of it can vote into it
    const int lx0 = cx0 - 1, ly0 = cy0 - 1;
    const int bx0 = imax(lx0 - max_r, 0) / EB, bx1 = imin(lx0 + VL - 1 + max_r, w - 1) / EB;
    const int by0 = imax(ly0 - max_r, 0) / EB, by1 = imin(ly0 + VL - 1 + max_r, h - 1) / EB;
    const int nbx = bx1 - bx0 + 1, nbin = nbx * (by1 - by0 + 1);
    // cells of this tile that exist in the image: one unsigned compare per axis covers "inside image" and "inside tile"
    const int vx_lo = imax(lx0, 0), vy_lo = imax(ly0, 0);
    const unsigned vx_n = (unsigned)(imin(lx0 + VL, w) - vx_lo), vy_n = (unsigned)(imin(ly0 + VL, h) - vy_lo);
    const int offx = vx_lo - lx0, offy = vy_lo - ly0;      // valid-cell origin inside the LDS tile (0 or 1)
    const int nsteps = max_r - min_r + 1;          // <= 31
    const int lane = tid & 63, wave = tid >> 6;
    const size_t bin_base = (size_t)bv * g.bins;
    // one wavefront per bin: a coalesced 512-byte load brings 64 edge records, every lane tests whether ITS record's
    // ray segment (+-max_r steps) can touch the tile at all, then the wave walks the surviving records one by one
    // (record broadcast through v_readlane, i.e. in scalar registers).  The reach window spans at most 7 x 7 bins; their
    // counts come from one lane-indexed load.  Bins are handed out dynamically (LDS ticket) because their populations
    // differ a lot (grid lines concentrate in a few bins): with a static split the waves of a workgroup spent 40 % of
    // their time waiting for the slowest one at the barrier.  Each wave keeps one bin in flight ahead of the one it walks.
    // lane q computes bin q's index once (the division by the window width is the expensive part); the walk fetches it
    // with v_readlane
    int my_cnt = 0, my_bin = 0;
    if (lane < nbin) {
        my_bin = (int)(bin_base + (size_t)(by0 + lane / nbx) * g.bw + (bx0 + lane % nbx));
        my_cnt = bin_cnt[my_bin];
    }
    int q = wave;                                   // first round: bin == wave index, later rounds: ticket
    int n_cur = 0;
    const uint2* ent_cur = bin_ent;
    uint2 mine = make_uint2(0u, 0u);
    if (q < nbin) {
        n_cur = __builtin_amdgcn_readlane(my_cnt, q);
        ent_cur = bin_ent + (size_t)__builtin_amdgcn_readlane(my_bin, q) * EB_CAP;
        if (lane < n_cur) mine = ent_cur[lane];
    }
    // The reach test is made per DIRECTION (a ray that points away from the tile would only occupy a lane for 30 steps):
    // surviving (record, direction) items are compacted into a per-wave LDS ring; whenever 64 are waiting they are walked
    // together, one item per lane, the radius steps as a loop with two adds per step.
    unsigned* ring = s_ring[wave];
    int fill = 0;
    while (q < nbin) {
        int qn = 0;
        if (lane == 0) qn = atomicAdd(&s_ticket, 1);
        qn = __builtin_amdgcn_readlane(qn, 0);
        int n_next = 0;
        const uint2* ent_next = bin_ent;
        uint2 mine_next = make_uint2(0u, 0u);
        if (qn < nbin) {
            n_next = __builtin_amdgcn_readlane(my_cnt, qn);
            ent_next = bin_ent + (size_t)__builtin_amdgcn_readlane(my_bin, qn) * EB_CAP;
            if (lane < n_next) mine_next = ent_next[lane];
        }
        const unsigned ent_base = (unsigned)(ent_cur - bin_ent);
        for (int k0 = 0; k0 < n_cur; k0 += 64) {
            if (k0 > 0) { mine = make_uint2(0u, 0u); if (k0 + lane < n_cur) mine = ent_cur[k0 + lane]; }
            bool reach_p = false, reach_n = false;
            if (k0 + lane < n_cur) {
                const int sxv = (int)(short)(mine.y & 0xffffu), syv = (int)(short)(mine.y >> 16);
                const int exr = (int)(mine.x & 0xffffu) - vx_lo, eyr = (int)(mine.x >> 16) - vy_lo;
                // the cells of direction +1 lie between the pixel and pixel + ((max_r * s) >> 10) on each axis (+-1 for the
                // floor), those of direction -1 between the pixel and pixel + ((-max_r * s) >> 10)
                // (24-bit multiplies: |s| <= 1024 and the radius is far below 2^23; v_mul_lo_u32 is quarter rate)
                const int px = __mul24(max_r, sxv), py = __mul24(max_r, syv);
                const int dxp = px >> 10, dyp = py >> 10;
                const int dxn = (-px) >> 10, dyn = (-py) >> 10;
                reach_p = exr + imax(dxp, 0) + 1 >= 0 && exr + imin(dxp, 0) - 1 < (int)vx_n &&
                          eyr + imax(dyp, 0) + 1 >= 0 && eyr + imin(dyp, 0) - 1 < (int)vy_n;
                reach_n = exr + imax(dxn, 0) + 1 >= 0 && exr + imin(dxn, 0) - 1 < (int)vx_n &&
                          eyr + imax(dyn, 0) + 1 >= 0 && eyr + imin(dyn, 0) - 1 < (int)vy_n;
            }
            const unsigned item = ent_base + (unsigned)(k0 + lane);
            const unsigned long long below = (1ull << lane) - 1ull;
            const unsigned long long mp = __ballot(reach_p);
            if (reach_p) ring[fill + __popcll(mp & below)] = item;
            fill += __popcll(mp);
            const unsigned long long mn = __ballot(reach_n);
            if (reach_n) ring[fill + __popcll(mn & below)] = item | 0x80000000u;
            fill += __popcll(mn);
            __builtin_amdgcn_wave_barrier();
            while (fill >= 64) {
                vote_walk64<NSTEPS>(ring, 64, lane, bin_ent, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
                // move the remainder (< 128 items) to the front
                const int rem = fill - 64;
                unsigned t0 = 0, t1 = 0;
                if (lane < rem) t0 = ring[64 + lane];
                if (64 + lane < rem) t1 = ring[128 + lane];
                __builtin_amdgcn_wave_barrier();
                if (lane < rem) ring[lane] = t0;
                if (64 + lane < rem) ring[64 + lane] = t1;
                fill = rem;
                __builtin_amdgcn_wave_barrier();
            }
        }
        q = qn; n_cur = n_next; ent_cur = ent_next; mine = mine_next;
    }
    // Every wavefront ends with fewer than 64 items in its ring, and a walk costs 30 steps whatever its fill: the eight remainders
    // are walked as ONE list, 64 items per wavefront (on a diagram 4 walks instead of 8 half-empty ones per tile, of ~28).  Lane l of
    // wavefront k takes item 64 k + l of the concatenated rings and parks it in the upper part of its own ring for the walk.
    if (lane == 0) s_fill[wave] = fill;
    __syncthreads();
    {
        int before = 0, r = 0, tot = 0;
        const int gi = wave * 64 + lane;
#pragma unroll
        for (int k = 0; k < VTHREADS / 64; k++) {
            const int f = s_fill[k];
            if (gi >= tot + f) { before = tot + f; r = k + 1; }
            tot += f;
        }
        const int cnt = imin(imax(tot - wave * 64, 0), 64);
        if (cnt > 0) {
            if (lane < cnt) ring[64 + lane] = s_ring[r][gi - before];
            __builtin_amdgcn_wave_barrier();
            vote_walk64<NSTEPS>(ring + 64, cnt, lane, bin_ent, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
        }
    }
    __syncthreads();
    // centre candidates: cells (x,y), 1 <= x <= w-1, 1 <= y <= h-1 (OpenCV scans padded rows 1..H, cols 1..W
    // of an accumulator whose votes sit at unpadded indices; cells x == W or y == H hold no votes).
    // tile cell (cx, cy): 16-bit half (cy >> 6) of dword (cy & 63) * VASTR + cx
#define I2S_CELL(cx, cy) ((int)((s_acc[((cy) & 63) * VASTR + (cx)] >> (((cy) >> 6) * 16)) & 0xffffu))
    if (dbg_acc) {
        for (int i = tid; i < VT * VT; i += VTHREADS) {
            const int ty = i / VT + 1, tx = i - (ty - 1) * VT + 1;
            const int x = lx0 + tx, y = ly0 + ty;
            if (x < w && y < h) dbg_acc[((size_t)bv * g.hmax + y) * g.pitch + x] = I2S_CELL(tx, ty);
        }
    }
    // one dword = cells (tx, r6) and (tx, r6 + 64): almost all hold fewer votes than the threshold and are rejected in pairs
    for (int i = tid; i < 64 * VT; i += VTHREADS) {
        const int r6 = i / VT, tx = i - r6 * VT + 1;
        const unsigned v2 = s_acc[r6 * VASTR + tx];
        if ((int)(v2 & 0xffffu) <= acc_thr && (int)(v2 >> 16) <= acc_thr) continue;
        const int x = lx0 + tx;
        if (x >= w || x < 1) continue;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int ty = r6 + 64 * hh;
            if (ty < 1 || ty > VT) continue;
            const int a = hh ? (int)(v2 >> 16) : (int)(v2 & 0xffffu);
            const int y = ly0 + ty;
            if (a <= acc_thr || y >= h || y < 1) continue;
            if (a > I2S_CELL(tx - 1, ty) && a >= I2S_CELL(tx + 1, ty) && a > I2S_CELL(tx, ty - 1) && a >= I2S_CELL(tx, ty + 1)) {
                const int k = atomicAdd(&cent_count[bv], 1);
                if (k < g.cent_cap) cent_list[(size_t)bv * g.cent_cap + k] = (unsigned)x | ((unsigned)y << 16);
            }
        }
    }
#undef I2S_CELL
}

// ---- round 4: two tiles per workgroup ------------------------------------------------------------------------------------
// The vote kernel is bound by the vector instructions of its radius walk (profiles/r03_b_vote_experiments.txt: 74 % of its
// instructions are walk steps, 10 vector + 4 scalar instructions each).  Two changes take 6 of the 10 out of most steps:
//  * the two 16-bit halves of an LDS dword are the same tile cell of TWO HoughCircles inputs (variants 2p and 2p + 1 of one image)
//    instead of two rows of one tile: the increment (1 or 0x10000) is a constant of the item -- it follows from which variant's
//    bin list the record came from -- and the row index needs no "& 63".  A workgroup of 1024 threads (16 wavefronts) owns the
//    pair: 128 x 129 dwords + 16 item rings = 78.5 KB, two workgroups per CU, the same 8 wavefronts per SIMD as before.
//  * a ray is a straight segment and the tile's valid cells are a box, so an item whose first (min_r) and last (max_r) cells are
//    both inside votes on EVERY step: such items (about six in ten) are collected apart and walked with no range test at all --
//    address (4 instructions), atomic, two adds.  The others keep the per-step test.
// Bit-exact by construction: the same (edge, direction, r) votes land in the same cells; only who walks them when has changed.
constexpr int VPT = 1024, VPW = VPT / 64;
#ifdef I2S_EXP_COUNT
__device__ unsigned long long g_vp_count[8];
#define VPC(i, n) do { if (lane == 0) atomicAdd(&g_vp_count[i], (unsigned long long)(n)); } while (0)
#else
#define VPC(i, n) do {} while (0)
#endif
#ifndef I2S_EXP_INLINE
#define I2S_EXP_INLINE __forceinline__
#endif
#ifndef I2S_EXP_FULL
#define I2S_EXP_FULL true
#endif

template <int NSTEPS, bool FULL>
__device__ I2S_EXP_INLINE void vote_walk_pair(unsigned item, bool active, const uint2* __restrict__ bin_ent, unsigned ent_split,
                                               int vx_lo, int vy_lo, unsigned vx_n, unsigned vy_n, int offx, int offy, int min_r,
                                               int nsteps, unsigned* __restrict__ s_acc)
{
    if (active) {
        const unsigned idx = item & 0x7fffffffu;
        const uint2 e = bin_ent[idx];                      // read a moment ago by the culling pass: an L1 / L2 hit
        int sx = (int)(short)(e.y & 0xffffu), sy = (int)(short)(e.y >> 16);
        if (item >> 31) { sx = -sx; sy = -sy; }
        // 22.10 fixed point relative to the LDS tile's first cell, as in vote_walk64
        int x = (((int)(e.x & 0xffffu) - vx_lo + offx) << 10) + __mul24(min_r, sx);
        int y = (((int)(e.x >> 16) - vy_lo + offy) << 10) + __mul24(min_r, sy);
        const unsigned val = idx >= ent_split ? 0x10000u : 1u;
        const unsigned xl = (vx_n + (unsigned)offx) << 10, yl = (vy_n + (unsigned)offy) << 10;
        auto step = [&]() {
            if (FULL || ((unsigned)x < xl && (unsigned)y < yl))
                atomicAdd(&s_acc[((unsigned)y >> 10) * (unsigned)VASTR + ((unsigned)x >> 10)], val);
            x += sx; y += sy;
        };
        if (NSTEPS > 0) {
#pragma unroll
            for (int st = 0; st < NSTEPS; st++) step();
        } else {
            for (int st = 0; st < nsteps; st++) step();
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// grid (tiles_x * tiles_y * nb * NVAR / 2), block 1024.  Same outputs as k_vote_centres.
template <int NSTEPS, bool SPLIT>
__global__ __launch_bounds__(VPT) void k_vote_pairs(const ImgDesc* __restrict__ desc, Geo g,
                                                    const uint2* __restrict__ bin_ent, const int* __restrict__ bin_cnt,
                                                    int min_r, int max_r, int acc_thr,
                                                    unsigned* __restrict__ cent_list, int* __restrict__ cent_count,
                                                    int* __restrict__ dbg_acc, int gx, int gy)
{
    __shared__ unsigned s_acc[VL * VASTR];         // cell (cx, cy): dword cy * VASTR + cx; low half variant 2p, high half variant 2p + 1
    __shared__ int s_ticket;
    __shared__ unsigned s_ring[VPW][VRING];        // per wavefront: always-inside items from the bottom, the others from the top
    __shared__ int s_fill[VPW][2];
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z / (NVAR / 2), v0 = (tl.z % (NVAR / 2)) * 2;
    const int w = desc[b].w, h = desc[b].h;
    const int cx0 = tl.tx * VT, cy0 = tl.ty * VT;    // first interior cell
    if (cx0 >= w || cy0 >= h) return;
    const int tid = threadIdx.x;
    const int bv = b * NVAR + v0;
    for (int i = tid; i < VL * VASTR; i += VPT) s_acc[i] = 0;
    if (tid == 0) s_ticket = VPW;                  // bins 0 .. 15 are taken by the waves' first round
    __syncthreads();
    const int lx0 = cx0 - 1, ly0 = cy0 - 1;
    const int bx0 = imax(lx0 - max_r, 0) / EB, bx1 = imin(lx0 + VL - 1 + max_r, w - 1) / EB;
    const int by0 = imax(ly0 - max_r, 0) / EB, by1 = imin(ly0 + VL - 1 + max_r, h - 1) / EB;
    const int nbx = bx1 - bx0 + 1, nbin = nbx * (by1 - by0 + 1);      // per variant, <= 49; bins nbin .. 2 nbin - 1 are variant 2p + 1's
    const int vx_lo = imax(lx0, 0), vy_lo = imax(ly0, 0);
    const unsigned vx_n = (unsigned)(imin(lx0 + VL, w) - vx_lo), vy_n = (unsigned)(imin(ly0 + VL, h) - vy_lo);
    const int offx = vx_lo - lx0, offy = vy_lo - ly0;
    const int nsteps = max_r - min_r + 1;
    const int lane = tid & 63, wave = tid >> 6;
    const unsigned ent_split = (unsigned)((size_t)(bv + 1) * g.bins * EB_CAP);     // first record index of variant 2p + 1
    const int xl = (int)((vx_n + (unsigned)offx) << 10), yl = (int)((vy_n + (unsigned)offy) << 10);
    int my_cnt_a = 0, my_cnt_b = 0, my_bin = 0;
    if (lane < nbin) {
        my_bin = (int)((size_t)bv * g.bins + (size_t)(by0 + lane / nbx) * g.bw + (bx0 + lane % nbx));
        my_cnt_a = bin_cnt[my_bin];
        my_cnt_b = bin_cnt[my_bin + g.bins];
    }
    auto bin_of = [&](int q, int& n, const uint2*& ent) {      // q is wave-uniform
        const int qq = q < nbin ? q : q - nbin;
        n = q < nbin ? __builtin_amdgcn_readlane(my_cnt_a, qq) : __builtin_amdgcn_readlane(my_cnt_b, qq);
        ent = bin_ent + ((size_t)__builtin_amdgcn_readlane(my_bin, qq) + (q < nbin ? 0 : (size_t)g.bins)) * EB_CAP;
    };
    int q = wave;
    int n_cur = 0;
    const uint2* ent_cur = bin_ent;
    uint2 mine = make_uint2(0u, 0u);
    if (q < 2 * nbin) {
        bin_of(q, n_cur, ent_cur);
        if (lane < n_cur) mine = ent_cur[lane];
    }
    unsigned* ring = s_ring[wave];
    int fill_f = 0, fill_p = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
    auto drain = [&]() {
        while (fill_f >= 64) {
            VPC(0, 1);
            vote_walk_pair<NSTEPS, I2S_EXP_FULL>(ring[lane], true, bin_ent, ent_split, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
            const int rem = fill_f - 64;               // < 64
            unsigned t0 = 0;
            if (lane < rem) t0 = ring[64 + lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < rem) ring[lane] = t0;
            fill_f = rem;
            __builtin_amdgcn_wave_barrier();
        }
        while (fill_p >= 64) {
            VPC(1, 1);
            vote_walk_pair<NSTEPS, false>(ring[VRING - 1 - lane], true, bin_ent, ent_split, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
            const int rem = fill_p - 64;
            unsigned t0 = 0;
            if (lane < rem) t0 = ring[VRING - 1 - 64 - lane];
            __builtin_amdgcn_wave_barrier();
            if (lane < rem) ring[VRING - 1 - lane] = t0;
            fill_p = rem;
            __builtin_amdgcn_wave_barrier();
        }
    };
    while (q < 2 * nbin) {
        int qn = 0;
        if (lane == 0) qn = atomicAdd(&s_ticket, 1);
        qn = __builtin_amdgcn_readlane(qn, 0);
        int n_next = 0;
        const uint2* ent_next = bin_ent;
        uint2 mine_next = make_uint2(0u, 0u);
        if (qn < 2 * nbin) {
            bin_of(qn, n_next, ent_next);
            if (lane < n_next) mine_next = ent_next[lane];
        }
        const unsigned ent_base = (unsigned)(ent_cur - bin_ent);
        for (int k0 = 0; k0 < n_cur; k0 += 64) {
            if (k0 > 0) { mine = make_uint2(0u, 0u); if (k0 + lane < n_cur) mine = ent_cur[k0 + lane]; }
            // Per direction d = +1 / -1 and axis the ray's coordinates run monotonically from X0 + d min_r s to X0 + d max_r s: the ray can
            // touch the tile only if that span meets [0, limit) on both axes (a superset: the per-step test decides), and it votes on
            // every step if both ends lie inside on both axes.
            bool in_p = false, in_n = false, all_p = false, all_n = false;
            if (k0 + lane < n_cur) {
                const int sxv = (int)(short)(mine.y & 0xffffu), syv = (int)(short)(mine.y >> 16);
                const int X0 = ((int)(mine.x & 0xffffu) - vx_lo + offx) << 10, Y0 = ((int)(mine.x >> 16) - vy_lo + offy) << 10;
                const int ax = __mul24(min_r, sxv), bx = __mul24(max_r, sxv), ay = __mul24(min_r, syv), by = __mul24(max_r, syv);
                const int mnx = imin(ax, bx), mxx = imax(ax, bx), mny = imin(ay, by), mxy = imax(ay, by);
                const int xlo_p = X0 + mnx, xhi_p = X0 + mxx, ylo_p = Y0 + mny, yhi_p = Y0 + mxy;
                const int xlo_n = X0 - mxx, xhi_n = X0 - mnx, ylo_n = Y0 - mxy, yhi_n = Y0 - mny;
                in_p = xhi_p >= 0 && xlo_p < xl && yhi_p >= 0 && ylo_p < yl;
                in_n = xhi_n >= 0 && xlo_n < xl && yhi_n >= 0 && ylo_n < yl;
                all_p = xlo_p >= 0 && xhi_p < xl && ylo_p >= 0 && yhi_p < yl;
                all_n = xlo_n >= 0 && xhi_n < xl && ylo_n >= 0 && yhi_n < yl;
            }
            const unsigned item = ent_base + (unsigned)(k0 + lane);
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const bool in = d ? in_n : in_p, all = SPLIT && (d ? all_n : all_p);
                const unsigned it = d ? (item | 0x80000000u) : item;
                const unsigned long long mf = __ballot(all), mp = __ballot(in && !all);
                if (all) ring[fill_f + __popcll(mf & below)] = it;
                else if (in) ring[VRING - 1 - (fill_p + __popcll(mp & below))] = it;
                fill_f += __popcll(mf); fill_p += __popcll(mp); VPC(4, __popcll(mf)); VPC(5, __popcll(mp)); VPC(6, 1);        // <= 63 + 63 + 64 entries: the two ends never meet
                __builtin_amdgcn_wave_barrier();
                drain();
            }
        }
        q = qn; n_cur = n_next; ent_cur = ent_next; mine = mine_next;
    }
    // leftovers (< 64 per class and wavefront): walked as two lists over the sixteen rings, 64 items per wavefront
    if (lane == 0) { s_fill[wave][0] = fill_f; s_fill[wave][1] = fill_p; }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 2; c++) {
        int before = 0, r = 0, tot = 0;
        const int gi = wave * 64 + lane;
#pragma unroll
        for (int k = 0; k < VPW; k++) {
            const int f = s_fill[k][c];
            if (gi >= tot + f) { before = tot + f; r = k + 1; }
            tot += f;
        }
        const int cnt = imin(imax(tot - wave * 64, 0), 64);
        if (cnt > 0) {
            VPC(2 + c, 1);
            unsigned it = 0;
            if (lane < cnt) it = c == 0 ? s_ring[r][gi - before] : s_ring[r][VRING - 1 - (gi - before)];
            if (c == 0) vote_walk_pair<NSTEPS, I2S_EXP_FULL>(it, lane < cnt, bin_ent, ent_split, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
            else vote_walk_pair<NSTEPS, false>(it, lane < cnt, bin_ent, ent_split, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
        }
    }
    __syncthreads();
    // centre candidates (as k_vote_centres): interior cells (tx, ty), 1 <= tx, ty <= VT; the two halves are two variants
#define I2S_CELLP(cx, cy, hh) ((int)((s_acc[(cy) * VASTR + (cx)] >> ((hh) * 16)) & 0xffffu))
    if (dbg_acc) {
        for (int i = tid; i < VT * VT; i += VPT) {
            const int ty = i / VT + 1, tx = i - (ty - 1) * VT + 1;
            const int x = lx0 + tx, y = ly0 + ty;
            if (x < w && y < h) {
                dbg_acc[((size_t)bv * g.hmax + y) * g.pitch + x] = I2S_CELLP(tx, ty, 0);
                dbg_acc[((size_t)(bv + 1) * g.hmax + y) * g.pitch + x] = I2S_CELLP(tx, ty, 1);
            }
        }
    }
    for (int i = tid; i < VT * VT; i += VPT) {
        const int ty = i / VT + 1, tx = i - (ty - 1) * VT + 1;
        const unsigned v2 = s_acc[ty * VASTR + tx];
        if ((int)(v2 & 0xffffu) <= acc_thr && (int)(v2 >> 16) <= acc_thr) continue;
        const int x = lx0 + tx, y = ly0 + ty;
        if (x >= w || x < 1 || y >= h || y < 1) continue;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int a = hh ? (int)(v2 >> 16) : (int)(v2 & 0xffffu);
            if (a <= acc_thr) continue;
            if (a > I2S_CELLP(tx - 1, ty, hh) && a >= I2S_CELLP(tx + 1, ty, hh) && a > I2S_CELLP(tx, ty - 1, hh) && a >= I2S_CELLP(tx, ty + 1, hh)) {
                const int k = atomicAdd(&cent_count[bv + hh], 1);
                if (k < g.cent_cap) cent_list[(size_t)(bv + hh) * g.cent_cap + k] = (unsigned)x | ((unsigned)y << 16);
            }
        }
    }
#undef I2S_CELLP
}

// ---- round 4b: ray lists -------------------------------------------------------------------------------------------------
// profiles/r04_a_vote_experiments.txt: 41 % of the vote kernel's vector instructions were the CULL -- every tile loaded and tested the
// records of all bins within reach, a record was tested by two to three tiles, and a load of 64 records left 38 of 128 (record,
// direction) candidates -- and more than half of its LDS cycles were bank conflicts, because the 64 items of a walk came from one
// 32 x 32 bin, i.e. one stone's outline, whose rays meet in one cell.  Both go away when the sorting is done ONCE, per record:
//   k_ray_lists : a workgroup per 128 x 32 pixel block (the block k_edge_bins wrote) computes, for each record and direction, the
//                 cell box of its ray and appends the item (record index | direction << 31) to the list of every accumulator tile the
//                 box meets (one or two per axis) -- in class F when the box lies inside the tile's valid cells (the ray then votes on
//                 every step: no range test in the walk), else in class P.  Appends are counted in LDS first: one global atomic per
//                 (workgroup, tile, class) and round of 128 records.
//   k_vote_lists: the pair kernel above without its cull: the 16 wavefronts take 64-item chunks of the tile's four lists (two
//                 variants x two classes) by ticket, and chunk c of a list of n chunks holds items c, c + n, c + 2 n ... -- the lanes
//                 of one walk are n items apart in the list, i.e. they belong to different stones.
// A list holds RL_CAP items; the reference's lists are unbounded, so a tile whose list overflowed (the count says so) is voted the old
// way, from the bins, which stay the ground truth (k_radius reads them too): the lists are an acceleration structure, never a limit.
constexpr int RL_CAP = 4096;           // items per (image, variant, tile, class)
constexpr int RLT = 64;                // k_ray_lists: ONE wavefront per 128 x 32 pixel block (no workgroup barrier anywhere)
constexpr int RL_R = 4;                // records per lane and batch: all loads of a batch are in flight together
#ifndef I2S_RL_WPB
#define I2S_RL_WPB 4
#endif
constexpr int RL_WPB = I2S_RL_WPB;     // wavefronts per workgroup
constexpr int RL_K = 8;                // blocks (of one row of bins) per wavefront
constexpr int RL_SLOTS = 12;           // lists a block can append to: 3 tile columns x 2 tile rows x 2 classes

// grid (ceil(bins_x / EBB_X) * ceil(bins_y / EBB_Y) * nb * NVAR) like k_edge_bins, block RLT.
// rl_cnt[((bv * g.vtiles + tile) * 2 + cls)], rl_items[... * RL_CAP + i]; cls 0 = F (the ray votes on every step), 1 = P.
//
// Per record and direction the cells of the ray span a box (monotone in r: the two ends).  On an axis the box meets the tiles
// ta = floor((lo - 1) / VT) .. tb = floor((hi + 1) / VT) (valid cells of tile t: t VT - 1 .. t VT + VT; a box is at most 31 cells long, so
// tb <= ta + 1), and ta == tb puts it strictly inside tile ta.  A ray with ta == tb on both axes and no cell outside the image is an F
// item of that one tile; every other ray is a P item of each of its (at most 2 x 2) tiles -- a superset of the tiles it votes in,
// which the walk's range test sorts out.  Three versions were measured (profiles/r04_a_vote_experiments.txt): a workgroup of 128
// threads counting with same-address LDS atomics between two barriers 5.2 us per diagram (a same-address LDS atomic costs ~4 cycles per
// lane, 32 of them per round), ballot loops over the list slots present 7.2 us (2 600 vector + 1 000 scalar instructions per 256 records);
// this one counts in PRIVATE LDS counters (slot x lane, conflict-free), turns them into offsets with one wave scan per slot in use and
// reserves the lists with one global atomic per slot and batch of 256 records.
__global__ __launch_bounds__(RLT * RL_WPB) void k_ray_lists(const ImgDesc* __restrict__ desc, Geo g, const uint2* __restrict__ bin_ent,
                                                   const int* __restrict__ bin_cnt, int min_r, int max_r,
                                                   unsigned* __restrict__ rl_items, int* __restrict__ rl_cnt, int gx, int gy, unsigned nunits)
{
    // A wavefront takes RL_K consecutive blocks of one row of bins (a whole row of a 1024-pixel image) and keeps the next batch's records
    // in flight while it sorts the current one: with one block per wavefront the kernel was a chain of three dependent memory round trips
    // per 200 records and took 4.3 us per diagram whatever the arithmetic between them (profiles/r04_a_vote_experiments.txt).
    const int gxk = (gx + RL_K - 1) / RL_K;
    const unsigned uix = tile_chunk_of_block(RL_WPB, nunits) + (threadIdx.x >> 6);
    if (uix >= nunits) return;
    const TileId tl = tile_of_index(uix, gxk, gy);
    const int b = tl.z / NVAR, v = tl.z % NVAR;
    const int w = desc[b].w, h = desc[b].h;
    const int bx_first = tl.tx * RL_K;                       // first block of the group
    const int y0 = tl.ty * (EBB_Y * EB);
    if (bx_first * (EBB_X * EB) >= w || y0 >= h) return;
    const int lane = threadIdx.x & 63;
    const int bv = b * NVAR + v;
    static_assert(EBB_Y == 1 && RL_K * EBB_X <= 64, "one row of bins per block; the group's bin counts fit one wavefront");
    const size_t row0 = (size_t)bv * g.bins + (size_t)tl.ty * g.bw + (size_t)bx_first * EBB_X;
    int cntv = 0;
    if (lane < RL_K * EBB_X && (bx_first * EBB_X + lane) * EB < w) cntv = bin_cnt[row0 + lane];
    const int ntx = (w + VT - 1) / VT, nty = (h + VT - 1) / VT;
    const int wty0 = imax(y0 - max_r - 1, 0) / VT;
    const int by1 = (wty0 + 1) * VT;
    auto block_total = [&](int i) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < EBB_X; k++) t += __builtin_amdgcn_readlane(cntv, i * EBB_X + k);
        return t;
    };
    auto load_batch = [&](int i, int r0, unsigned (&idx)[RL_R], uint2 (&e)[RL_R]) {
        int n[EBB_X], ntot = 0;
#pragma unroll
        for (int k = 0; k < EBB_X; k++) { n[k] = __builtin_amdgcn_readlane(cntv, i * EBB_X + k); ntot += n[k]; }
        const size_t bin0 = row0 + (size_t)i * EBB_X;
#pragma unroll
        for (int j = 0; j < RL_R; j++) {
            const int ri = r0 + j * RLT + lane;
            int k = 0, off = ri;
#pragma unroll
            for (int q = 0; q < EBB_X - 1; q++) if (k == q && off >= n[q]) { off -= n[q]; k = q + 1; }
            idx[j] = (unsigned)((bin0 + k) * EB_CAP + off);
            e[j] = make_uint2(0u, 0u);
            if (ri < ntot) e[j] = bin_ent[idx[j]];
        }
    };
    auto sort_batch = [&](int i, int r0, const unsigned (&idx)[RL_R], const uint2 (&e)[RL_R]) {
        const int ntot = block_total(i);
        // tiles the block's rays can meet: columns wtx0 .. wtx0 + 2, rows wty0 .. wty0 + 1 (a ray moves at most max_r <= 30 cells per
        // axis); a lane finds a tile index by comparing with the (uniform) first cells of the next columns / row.  Tile t = row * 3 + column.
        const int x0 = (bx_first + i) * (EBB_X * EB);
        const int wtx0 = imax(x0 - max_r - 1, 0) / VT;
        const int bx1 = (wtx0 + 1) * VT, bx2 = (wtx0 + 2) * VT;
        const size_t list0 = (size_t)bv * g.vtiles + (size_t)wty0 * g.vtx + wtx0;
        // per (record, direction): first tile t0 | F << 3 | second column << 4 | second row << 5, or -1; and the lane's item counts per
        // tile in 5-bit fields (at most 8 rays per lane and batch), F and P apart
        int code[RL_R][2];
        unsigned cnt_f = 0, cnt_p = 0;
#pragma unroll
        for (int j = 0; j < RL_R; j++) {
            const bool valid = r0 + j * RLT + lane < ntot;
            const int sx = (int)(short)(e[j].y & 0xffffu), sy = (int)(short)(e[j].y >> 16);
            const int ex = (int)(e[j].x & 0xffffu), ey = (int)(e[j].x >> 16);
            const int ax = __mul24(min_r, sx), bx = __mul24(max_r, sx), ay = __mul24(min_r, sy), by = __mul24(max_r, sy);
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const int x1 = ex + ((d ? -ax : ax) >> 10), x2 = ex + ((d ? -bx : bx) >> 10);
                const int y1 = ey + ((d ? -ay : ay) >> 10), y2 = ey + ((d ? -by : by) >> 10);
                const int xlo = imin(x1, x2), xhi = imax(x1, x2), ylo = imin(y1, y2), yhi = imax(y1, y2);
                const int ta = (xlo - 1 >= bx1) + (xlo - 1 >= bx2), tb = imin((xhi + 1 >= bx1) + (xhi + 1 >= bx2), ntx - 1 - wtx0);
                const int ua = (ylo - 1 >= by1), ub = imin((yhi + 1 >= by1), nty - 1 - wty0);
                const bool outside = xhi < 0 || xlo >= w || yhi < 0 || ylo >= h;
                const bool clipped = xlo < 0 || xhi >= w || ylo < 0 || yhi >= h;
                const int twox = tb > ta, twoy = ub > ua;
                const bool full = !twox && !twoy && !clipped;
                int cd = -1;
                if (valid && !outside) {
                    const int t0 = ua * 3 + ta;
                    cd = t0 | (full ? 8 : 0) | (twox << 4) | (twoy << 5);
                    // P items: tiles t0, t0 + 1 (second column), t0 + 3 (second row), t0 + 4 (both)
                    if (full) cnt_f += 1u << (5 * t0);
                    else cnt_p += (1u + ((unsigned)twox << 5) + ((unsigned)twoy << 15) + ((unsigned)(twox & twoy) << 20)) << (5 * t0);
                }
                code[j][d] = cd;
            }
        }
        // counts -> offsets: inclusive scan over the lanes, per tile (F count | P count << 16); the last lane holds the totals
        int own[6], inc[6];
#pragma unroll
        for (int t = 0; t < 6; t++) own[t] = inc[t] = (int)(((cnt_f >> (5 * t)) & 31u) | (((cnt_p >> (5 * t)) & 31u) << 16));
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            int up[6];
#pragma unroll
            for (int t = 0; t < 6; t++) up[t] = __shfl_up(inc[t], (unsigned)o);
#pragma unroll
            for (int t = 0; t < 6; t++) if (lane >= o) inc[t] += up[t];
        }
        int tot[6];
#pragma unroll
        for (int t = 0; t < 6; t++) tot[t] = __builtin_amdgcn_readlane(inc[t], 63);
        // lane t reserves tile t's two lists with ONE 64-bit atomic (the F and the P counter are neighbours)
        int base_f = 0, base_p = 0;
        {
            int mine = 0;
#pragma unroll
            for (int t = 0; t < 6; t++) if (lane == t) mine = tot[t];
            if (lane < 6 && mine != 0) {
                const size_t li = list0 + (size_t)(lane / 3) * g.vtx + (lane % 3);
                const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(rl_cnt) + li,
                                                         (unsigned long long)(mine & 0xffff) | ((unsigned long long)(mine >> 16) << 32));
                base_f = (int)(unsigned)old; base_p = (int)(unsigned)(old >> 32);
            }
        }
#pragma unroll
        for (int t = 0; t < 6; t++) {
            if (tot[t] == 0) continue;                                     // uniform
            const int bf = __builtin_amdgcn_readlane(base_f, t), bp = __builtin_amdgcn_readlane(base_p, t);
            unsigned* lf = rl_items + ((list0 + (size_t)(t / 3) * g.vtx + (t % 3)) * 2) * RL_CAP;
            unsigned* lp = lf + RL_CAP;
            int cur_f = bf + ((inc[t] - own[t]) & 0xffff), cur_p = bp + ((inc[t] - own[t]) >> 16);
#pragma unroll
            for (int j = 0; j < RL_R; j++) {
#pragma unroll
                for (int d = 0; d < 2; d++) {
                    const int cd = code[j][d];
                    if (cd < 0) continue;
                    const int t0 = cd & 7;
                    const bool hit = t == t0 || ((cd & 16) && t == t0 + 1) || ((cd & 32) && t == t0 + 3) || ((cd & 48) == 48 && t == t0 + 4);
                    if (!hit) continue;
                    const unsigned item = idx[j] | (d ? 0x80000000u : 0u);
                    if (cd & 8) { if (cur_f < RL_CAP) lf[cur_f] = item; cur_f++; }
                    else { if (cur_p < RL_CAP) lp[cur_p] = item; cur_p++; }
                }
            }
        }
    };
    int ci = 0;
    while (ci < RL_K && block_total(ci) == 0) ci++;
    if (ci >= RL_K) return;
    int cr0 = 0;
    unsigned idx_a[RL_R], idx_b[RL_R];
    uint2 e_a[RL_R], e_b[RL_R];
    load_batch(ci, 0, idx_a, e_a);
    while (ci < RL_K) {
        int ni = ci, nr0 = cr0 + RLT * RL_R;
        if (nr0 >= block_total(ci)) { nr0 = 0; do ni++; while (ni < RL_K && block_total(ni) == 0); }
        if (ni < RL_K) load_batch(ni, nr0, idx_b, e_b);
        sort_batch(ci, cr0, idx_a, e_a);
        ci = ni; cr0 = nr0;
#pragma unroll
        for (int j = 0; j < RL_R; j++) { idx_a[j] = idx_b[j]; e_a[j] = e_b[j]; }
    }
}

// Walk of one item with the increment given (see vote_walk_pair).
template <int NSTEPS, bool FULL>
__device__ __forceinline__ void vote_walk_item(unsigned item, bool active, unsigned val, const uint2* __restrict__ bin_ent,
                                               int vx_lo, int vy_lo, unsigned vx_n, unsigned vy_n, int offx, int offy, int min_r,
                                               int nsteps, unsigned* __restrict__ s_acc)
{
    if (active) {
        const uint2 e = bin_ent[item & 0x7fffffffu];
        int sx = (int)(short)(e.y & 0xffffu), sy = (int)(short)(e.y >> 16);
        if (item >> 31) { sx = -sx; sy = -sy; }
        int x = (((int)(e.x & 0xffffu) - vx_lo + offx) << 10) + __mul24(min_r, sx);
        int y = (((int)(e.x >> 16) - vy_lo + offy) << 10) + __mul24(min_r, sy);
        const unsigned xl = (vx_n + (unsigned)offx) << 10, yl = (vy_n + (unsigned)offy) << 10;
        auto step = [&]() {
            if (FULL || ((unsigned)x < xl && (unsigned)y < yl))
                atomicAdd(&s_acc[((unsigned)y >> 10) * (unsigned)VASTR + ((unsigned)x >> 10)], val);
            x += sx; y += sy;
        };
        if (NSTEPS > 0) {
#pragma unroll
            for (int st = 0; st < NSTEPS; st++) step();
        } else {
            for (int st = 0; st < nsteps; st++) step();
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// grid (tiles_x * tiles_y * nb * NVAR / 2), block 1024.  Same outputs as k_vote_centres.
template <int NSTEPS>
__global__ __launch_bounds__(VPT) void k_vote_lists(const ImgDesc* __restrict__ desc, Geo g,
                                                    const uint2* __restrict__ bin_ent, const int* __restrict__ bin_cnt,
                                                    const unsigned* __restrict__ rl_items, const int* __restrict__ rl_cnt,
                                                    int min_r, int max_r, int acc_thr,
                                                    unsigned* __restrict__ cent_list, int* __restrict__ cent_count,
                                                    int* __restrict__ dbg_acc, int gx, int gy)
{
    __shared__ unsigned s_acc[VL * VASTR];         // cell (cx, cy): dword cy * VASTR + cx; low half variant 2p, high half variant 2p + 1
    __shared__ int s_ticket;
    __shared__ unsigned s_ring[VPW][VRING];        // overflowed tiles only
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z / (NVAR / 2), v0 = (tl.z % (NVAR / 2)) * 2;
    const int w = desc[b].w, h = desc[b].h;
    const int cx0 = tl.tx * VT, cy0 = tl.ty * VT;
    if (cx0 >= w || cy0 >= h) return;
    const int tid = threadIdx.x;
    const int bv = b * NVAR + v0;
    for (int i = tid; i < VL * VASTR; i += VPT) s_acc[i] = 0;
    if (tid == 0) s_ticket = VPW;                  // chunks 0 .. 15 are taken by the waves' first round
    const int lx0 = cx0 - 1, ly0 = cy0 - 1;
    const int vx_lo = imax(lx0, 0), vy_lo = imax(ly0, 0);
    const unsigned vx_n = (unsigned)(imin(lx0 + VL, w) - vx_lo), vy_n = (unsigned)(imin(ly0 + VL, h) - vy_lo);
    const int offx = vx_lo - lx0, offy = vy_lo - ly0;
    const int nsteps = max_r - min_r + 1;
    const int lane = tid & 63, wave = tid >> 6;
    // the four lists of this tile pair: segment s = 2 * variant + class
    const size_t l0 = ((size_t)bv * g.vtiles + (size_t)tl.ty * g.vtx + tl.tx) * 2;
    int cnt[4], nch[4], first[5];
    bool over[2];
#pragma unroll
    for (int hv = 0; hv < 2; hv++) {
        cnt[2 * hv] = rl_cnt[l0 + (size_t)hv * g.vtiles * 2];
        cnt[2 * hv + 1] = rl_cnt[l0 + (size_t)hv * g.vtiles * 2 + 1];
        over[hv] = cnt[2 * hv] > RL_CAP || cnt[2 * hv + 1] > RL_CAP;
    }
    first[0] = 0;
#pragma unroll
    for (int sg = 0; sg < 4; sg++) { nch[sg] = over[sg >> 1] ? 0 : (cnt[sg] + 63) >> 6; first[sg + 1] = first[sg] + nch[sg]; }
    __syncthreads();
    for (int c = wave; c < first[4]; ) {
        int sg = 0;
#pragma unroll
        for (int k = 1; k < 4; k++) if (c >= first[k]) sg = k;
        // chunk cc of a list of nch[sg] chunks takes items cc, cc + nch, cc + 2 nch, ...: one item per lane
#ifdef I2S_EXP_NOTRANSPOSE
        const int i = (c - first[sg]) * 64 + lane;
#else
        const int i = lane * nch[sg] + (c - first[sg]);
#endif
        const bool act = i < cnt[sg];
        unsigned item = 0;
        if (act) item = rl_items[(l0 + (size_t)(sg >> 1) * g.vtiles * 2 + (sg & 1)) * RL_CAP + i];
        const unsigned val = (sg >> 1) ? 0x10000u : 1u;
        if (sg & 1) vote_walk_item<NSTEPS, false>(item, act, val, bin_ent, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
        else vote_walk_item<NSTEPS, true>(item, act, val, bin_ent, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
        int cn = 0;
        if (lane == 0) cn = atomicAdd(&s_ticket, 1);
        c = __builtin_amdgcn_readlane(cn, 0);
    }
    // a variant whose list overflowed: tile-major cull of its bins, as k_vote_pairs does it (rolled walk: this is the rare path)
#pragma unroll 1
    for (int hv = 0; hv < 2; hv++) {
        if (!over[hv]) continue;                   // uniform over the workgroup
        __syncthreads();
        if (tid == 0) s_ticket = VPW;
        __syncthreads();
        const int bx0 = imax(lx0 - max_r, 0) / EB, bx1 = imin(lx0 + VL - 1 + max_r, w - 1) / EB;
        const int by0 = imax(ly0 - max_r, 0) / EB, by1 = imin(ly0 + VL - 1 + max_r, h - 1) / EB;
        const int nbx = bx1 - bx0 + 1, nbin = nbx * (by1 - by0 + 1);
        const int xl = (int)((vx_n + (unsigned)offx) << 10), yl = (int)((vy_n + (unsigned)offy) << 10);
        const unsigned val = hv ? 0x10000u : 1u;
        int my_cnt = 0, my_bin = 0;
        if (lane < nbin) {
            my_bin = (int)((size_t)(bv + hv) * g.bins + (size_t)(by0 + lane / nbx) * g.bw + (bx0 + lane % nbx));
            my_cnt = bin_cnt[my_bin];
        }
        unsigned* ring = s_ring[wave];
        int fill = 0;
        const unsigned long long below = (1ull << lane) - 1ull;
        for (int q = wave; q < nbin; ) {
            const int n_cur = __builtin_amdgcn_readlane(my_cnt, q);
            const uint2* ent_cur = bin_ent + (size_t)__builtin_amdgcn_readlane(my_bin, q) * EB_CAP;
            const unsigned ent_base = (unsigned)(ent_cur - bin_ent);
            for (int k0 = 0; k0 < n_cur; k0 += 64) {
                bool in_p = false, in_n = false;
                if (k0 + lane < n_cur) {
                    const uint2 mine = ent_cur[k0 + lane];
                    const int sxv = (int)(short)(mine.y & 0xffffu), syv = (int)(short)(mine.y >> 16);
                    const int X0 = ((int)(mine.x & 0xffffu) - vx_lo + offx) << 10, Y0 = ((int)(mine.x >> 16) - vy_lo + offy) << 10;
                    const int ax = __mul24(min_r, sxv), bx = __mul24(max_r, sxv), ay = __mul24(min_r, syv), by = __mul24(max_r, syv);
                    const int mnx = imin(ax, bx), mxx = imax(ax, bx), mny = imin(ay, by), mxy = imax(ay, by);
                    in_p = X0 + mxx >= 0 && X0 + mnx < xl && Y0 + mxy >= 0 && Y0 + mny < yl;
                    in_n = X0 - mnx >= 0 && X0 - mxx < xl && Y0 - mny >= 0 && Y0 - mxy < yl;
                }
                const unsigned item = ent_base + (unsigned)(k0 + lane);
                const unsigned long long mp = __ballot(in_p);
                if (in_p) ring[fill + __popcll(mp & below)] = item;
                fill += __popcll(mp);
                const unsigned long long mn = __ballot(in_n);
                if (in_n) ring[fill + __popcll(mn & below)] = item | 0x80000000u;
                fill += __popcll(mn);
                __builtin_amdgcn_wave_barrier();
                while (fill >= 64) {
                    vote_walk_item<0, false>(ring[lane], true, val, bin_ent, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
                    const int rem = fill - 64;
                    unsigned t0 = 0, t1 = 0;
                    if (lane < rem) t0 = ring[64 + lane];
                    if (64 + lane < rem) t1 = ring[128 + lane];
                    __builtin_amdgcn_wave_barrier();
                    if (lane < rem) ring[lane] = t0;
                    if (64 + lane < rem) ring[64 + lane] = t1;
                    fill = rem;
                    __builtin_amdgcn_wave_barrier();
                }
            }
            int qn = 0;
            if (lane == 0) qn = atomicAdd(&s_ticket, 1);
            q = __builtin_amdgcn_readlane(qn, 0);
        }
        vote_walk_item<0, false>(lane < fill ? ring[lane] : 0u, lane < fill, val, bin_ent, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
    }
    __syncthreads();
    // centre candidates: interior cells (tx, ty), 1 <= tx, ty <= VT; the two halves are two variants
#define I2S_CELLP(cx, cy, hh) ((int)((s_acc[(cy) * VASTR + (cx)] >> ((hh) * 16)) & 0xffffu))
    if (dbg_acc) {
        for (int i = tid; i < VT * VT; i += VPT) {
            const int ty = i / VT + 1, tx = i - (ty - 1) * VT + 1;
            const int x = lx0 + tx, y = ly0 + ty;
            if (x < w && y < h) {
                dbg_acc[((size_t)bv * g.hmax + y) * g.pitch + x] = I2S_CELLP(tx, ty, 0);
                dbg_acc[((size_t)(bv + 1) * g.hmax + y) * g.pitch + x] = I2S_CELLP(tx, ty, 1);
            }
        }
    }
    for (int i = tid; i < VT * VT; i += VPT) {
        const int ty = i / VT + 1, tx = i - (ty - 1) * VT + 1;
        const unsigned v2 = s_acc[ty * VASTR + tx];
        if ((int)(v2 & 0xffffu) <= acc_thr && (int)(v2 >> 16) <= acc_thr) continue;
        const int x = lx0 + tx, y = ly0 + ty;
        if (x >= w || x < 1 || y >= h || y < 1) continue;
#pragma unroll
        for (int hh = 0; hh < 2; hh++) {
            const int a = hh ? (int)(v2 >> 16) : (int)(v2 & 0xffffu);
            if (a <= acc_thr) continue;
            if (a > I2S_CELLP(tx - 1, ty, hh) && a >= I2S_CELLP(tx + 1, ty, hh) && a > I2S_CELLP(tx, ty - 1, hh) && a >= I2S_CELLP(tx, ty + 1, hh)) {
                const int k = atomicAdd(&cent_count[bv + hh], 1);
                if (k < g.cent_cap) cent_list[(size_t)(bv + hh) * g.cent_cap + k] = (unsigned)x | ((unsigned)y << 16);
            }
        }
    }
#undef I2S_CELLP
}

// Sort key of an estimated circle; ascending key order == OpenCV's cmpAccum order
// (accum desc, radius desc, x asc, y asc).  s = upbin + j of the radius histogram scan (radius = s/20 + min_r).
__device__ __forceinline__ unsigned long long est_key(int acc, int s, int x, int y)
{
    return ((unsigned long long)(4095 - acc) << 42) | ((unsigned long long)(1023 - s) << 32) |
           ((unsigned long long)(unsigned)x << 16) | (unsigned long long)(unsigned)y;
}

constexpr int RAD_BINS_MAX = 320;   // 5 x 64
constexpr int RAD_LUT = 512;        // radius-bin table: K / 2 < max_r^2 / 2 <= 450 (max_r <= 30, checked by the host)

// Radius estimate + support check of every centre candidate (hough.cpp HoughCircleEstimateRadiusInvoker).
// grid (RAD_GX, nb * NVAR), block 256 = 4 independent wavefronts, one centre per wavefront at a time.
// The voting edge pixels near the centre come from the edge bins (the same set OpenCV keeps in `nz`); their distances
// go into a 10-bins-per-pixel LDS histogram; the histogram scan (windows of 10 bins opened at every non-empty bin,
// walking down from the largest radius) runs wave-uniformly on prefix sums + 64-bit occupancy masks.
// est_keys[(b*NVAR+v) * g.est_cap + i], est_count[b*NVAR+v].
__global__ __launch_bounds__(256) void k_radius(const ImgDesc* __restrict__ desc, Geo g,
                                                const uint2* __restrict__ bin_ent, const int* __restrict__ bin_cnt,
                                                const unsigned* __restrict__ cent_list, const int* __restrict__ cent_count,
                                                int min_r, int max_r, int acc_thr,
                                                unsigned long long* __restrict__ est_keys, int* __restrict__ est_count)
{
    __shared__ int s_bins[4][RAD_BINS_MAX];
    __shared__ unsigned short s_lut[RAD_LUT];
    const int bv = blockIdx.y;
    const int b = bv / NVAR;
    const int w = desc[b].w, h = desc[b].h;
    const int n = imin(cent_count[bv], g.cent_cap);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nBinsPerDr = 10;
    int nBins = __float2int_rn((float)(max_r - min_r) / 1.0f * (float)nBinsPerDr);
    if (nBins < 1) nBins = 1;
    if (blockIdx.x * 4 >= n) return;
    // The centre is (cxi + 0.5, cyi + 0.5) and edge pixels are integers, so the squared distance OpenCV computes in float is
    // exactly K + 0.5 with K = dx (dx + 1) + dy (dy + 1), an even integer: the radius bin is a function of K / 2 alone.  The
    // table holds OpenCV's own float expression evaluated once per K; minR^2 <= K + 0.5 <= maxR^2  <=>  min_r^2 <= K < max_r^2.
    const int k_lo = min_r * min_r, k_hi = max_r * max_r;
    for (int i = threadIdx.x; i < RAD_LUT; i += 256) {
        const float d = sqrtf((float)(2 * i) + 0.5f);
        const int bi = __float2int_rn((d - (float)min_r) / 1.0f * (float)nBinsPerDr);
        s_lut[i] = (unsigned short)imax(0, imin(nBins - 1, bi));
    }
    __syncthreads();
    const size_t bin_base = (size_t)bv * g.bins;
    int* bins = s_bins[wave];
    // every wavefront works through its centres on its own: the phases below are separated by wave-level barriers only
    // (LDS operations of one wave complete in order), so the four waves of a workgroup never wait for each other.
    // The kernel is bound by load latency (centre -> bin counts -> records, one after the other, per centre), so the loads are
    // pipelined across centres: while centre c is processed, the bin counts and the first 64 slots of every bin of centre c + 1
    // are in flight (the slots are fetched WITHOUT knowing the counts: stale bytes past a bin's count are masked later), and the
    // coordinates of centre c + 2 are on their way.
    const int stride = gridDim.x * 4;
    const unsigned* clist = cent_list + (size_t)bv * g.cent_cap;
    struct Win { int nbx, nbin, my_cnt, my_bin; unsigned pre[9]; };
    auto fetch = [&](unsigned e, Win& W) {
        const int cxi = (int)(e & 0xffffu), cyi = (int)(e >> 16);
        // pixels with minR^2 <= d^2 <= maxR^2 lie within max_r of the centre: at most 3 x 3 bins overlap that box
        const int bx0 = imax(cxi - max_r, 0) / EB, bx1 = imin(cxi + max_r + 1, w - 1) / EB;
        const int by0 = imax(cyi - max_r, 0) / EB, by1 = imin(cyi + max_r + 1, h - 1) / EB;
        W.nbx = bx1 - bx0 + 1; W.nbin = W.nbx * (by1 - by0 + 1);          // <= 9
        W.my_cnt = 0; W.my_bin = 0;
        if (lane < W.nbin) {
            W.my_bin = (int)(bin_base + (size_t)(by0 + lane / W.nbx) * g.bw + (bx0 + lane % W.nbx));
            W.my_cnt = bin_cnt[W.my_bin];
        }
#pragma unroll
        for (int q = 0; q < 9; q++) {
            W.pre[q] = 0xffffffffu;
            if (q < W.nbin) W.pre[q] = bin_ent[(size_t)__builtin_amdgcn_readlane(W.my_bin, q) * EB_CAP + lane].x;
        }
    };
    int c = blockIdx.x * 4 + wave;
    unsigned e_cur = c < n ? clist[c] : 0u, e_next = c + stride < n ? clist[c + stride] : 0u;
    Win cur, nxt;
    if (c < n) fetch(e_cur, cur);
    for (; c < n; c += stride) {
        const unsigned e_next2 = c + 2 * stride < n ? clist[c + 2 * stride] : 0u;
        if (c + stride < n) fetch(e_next, nxt);
        for (int i = lane; i < RAD_BINS_MAX; i += 64) bins[i] = 0;
        __builtin_amdgcn_wave_barrier();
        const int cxi = (int)(e_cur & 0xffffu), cyi = (int)(e_cur >> 16);
        {
#pragma unroll
            for (int q = 0; q < 9; q++) {
                if (q >= cur.nbin) continue;
                const int cnt_q = __builtin_amdgcn_readlane(cur.my_cnt, q);
                const uint2* ent = bin_ent + (size_t)__builtin_amdgcn_readlane(cur.my_bin, q) * EB_CAP;
                for (int k = lane; k < cnt_q; k += 64) {
                    const unsigned xy = k < 64 ? cur.pre[q] : ent[k].x;
                    const int dxi = cxi - (int)(xy & 0xffffu), dyi = cyi - (int)(xy >> 16);
                    const int K = __mul24(dxi, dxi + 1) + __mul24(dyi, dyi + 1);
                    if (K >= k_lo && K < k_hi) atomicAdd(&bins[s_lut[K >> 1]], 1);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // occupancy masks (64 bins per ballot), then inclusive prefix sums P[i] in place: every lane owns 5 consecutive bins, so
        // one wave scan of the per-lane totals suffices
        unsigned long long occ[RAD_BINS_MAX / 64];
#pragma unroll
        for (int q = 0; q < RAD_BINS_MAX / 64; q++) occ[q] = __ballot(bins[q * 64 + lane] != 0);
        __builtin_amdgcn_wave_barrier();
        constexpr int PERL = RAD_BINS_MAX / 64;
        int loc[PERL];
#pragma unroll
        for (int q = 0; q < PERL; q++) loc[q] = bins[lane * PERL + q] + (q ? loc[q - 1] : 0);
        int vsum = loc[PERL - 1];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl(vsum, lane >= d ? lane - d : lane);
            if (lane >= d) vsum += t;
        }
        const int excl = vsum - loc[PERL - 1];
#pragma unroll
        for (int q = 0; q < PERL; q++) bins[lane * PERL + q] = loc[q] + excl;
        __builtin_amdgcn_wave_barrier();
        {
            // OpenCV scans the histogram from the top: take the highest non-empty bin u <= j, sum the 10 bins below it
            // (lo = u - 10), compare, continue at j = lo - 1.  WHICH bins get visited depends on the occupancy masks alone, so
            // that part runs as a scalar bit loop, one statically indexed 64-bin word after the other (a dynamically indexed
            // mask array costs a select chain per step on the scalar unit the whole CU shares), and hands visited bin k to
            // lane k; the window sums and radii of all visited bins are then formed in parallel (one LDS round trip, one
            // division), and only the order-dependent comparison chain is folded sequentially, reading lane k's candidate with
            // v_readlane.
            int u_mine = 0, nv = 0;
            int j = nBins - 1;
#pragma unroll
            for (int q = RAD_BINS_MAX / 64 - 1; q >= 0; q--) {
                const unsigned long long wq = q == 0 ? (occ[0] & ~1ull) : occ[q];       // a visited bin has u >= 1
                while (j > 0 && j >= q * 64) {
                    const int top = imin(j - q * 64, 63);
                    const unsigned long long m = top < 63 ? (wq & ((2ull << top) - 1ull)) : wq;
                    if (!m) break;                                   // nothing at or below j in this word: go on in the next one
                    const int u = q * 64 + 63 - __clzll((long long)m);
                    if (lane == nv) u_mine = u;
                    nv++;
                    j = imax(u - nBinsPerDr, -1) - 1;               // lo - 1: the outer loop's own j--
                }
            }
            int c_cnt = 0, c_s = 0;
            float c_r = 0.f;
            if (lane < nv) {
                const int lo = imax(u_mine - nBinsPerDr, -1);       // j after OpenCV's inner summing loop
                c_cnt = bins[u_mine] - (lo >= 0 ? bins[lo] : 0);
                c_r = (float)(u_mine + lo) / 2.f / (float)nBinsPerDr * 1.0f + (float)min_r;
                c_s = u_mine + lo;
            }
            int maxCount = 0, sBest = 0;
            float rBest = 0.f;
            for (int k = 0; k < nv; k++) {
                const int curCount = __builtin_amdgcn_readlane(c_cnt, k);
                const float rCur = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(c_r), k));
                if (((float)curCount * rBest >= (float)maxCount * rCur) || (rBest < 1.1920929e-07f && curCount >= maxCount)) {
                    rBest = rCur; maxCount = curCount; sBest = __builtin_amdgcn_readlane(c_s, k);
                }
            }
            if (lane == 0 && maxCount > acc_thr) {
                const int k = atomicAdd(&est_count[bv], 1);
                if (k < g.est_cap) est_keys[(size_t)bv * g.est_cap + k] = est_key(imin(maxCount, 4095), sBest, cxi, cyi);
            }
        }
        __builtin_amdgcn_wave_barrier();
        e_cur = e_next; e_next = e_next2; cur = nxt;
    }
}

constexpr int FIN_THREADS = 1024;
constexpr int FIN_BUCKETS = 4096;      // hash buckets of the RemoveOverlaps neighbour grid (HASH instantiation)

// grid (nb * NVAR), block FIN_THREADS.  Sorts the estimates (OpenCV's cmpAccum order), runs RemoveOverlaps (keep a circle
// iff it is at least min_dist from every circle already kept) and writes circles (x, y, r) in output order.
// vcirc[(bv * g.vcirc_cap + i) * 3], vcount[bv]; overflow[b] is set when a capacity was exceeded.
// ECAP / VCAP: compile-time capacities of the LDS arrays (>= g.est_cap / g.vcirc_cap); the host launches the instantiation that
// fits the context's capacities, so that 1024 x 1024 contexts keep the small footprint (several workgroups per CU).
// HASH: candidates are also chained into a grid of min_dist-sized cells (hashed into FIN_BUCKETS LDS list heads), so that
// RemoveOverlaps looks at the 3 x 3 cells around a candidate instead of at every earlier one: the Gaussian variants of a diagram
// yield ~750 estimates, and the all-pairs sweep was 70 % of this kernel's time.  The large-capacity instantiation has no LDS
// left for the grid and keeps the sweep.
template <int ECAP, int VCAP, bool HASH>
__global__ __launch_bounds__(1024) void k_circles_final(Geo g, const unsigned long long* __restrict__ est_keys,
                                                       const int* __restrict__ est_count, const int* __restrict__ cent_count,
                                                       float min_dist, int min_r,
                                                       float* __restrict__ vcirc, int* __restrict__ vcount, int* __restrict__ overflow)
{
    // gfx950 only: the large-capacity instantiation (ECAP = 16384: 128 KB of keys + 16 KB of status bytes) needs the 160 KB of LDS a
    // CDNA4 CU has, runs one workgroup per CU and keeps the all-pairs sweep (no LDS left for the neighbour grid): contexts of more
    // than one megapixel trade speed on crowded images for capacity.
    static_assert(sizeof(unsigned long long) * ECAP + sizeof(short) * VCAP + (HASH ? 4 * FIN_BUCKETS + 2 * ECAP : 6) + 128 <= 160 * 1024,
                  "k_circles_final: LDS arrays exceed a gfx950 CU's 160 KB");
    __shared__ unsigned long long s_key[ECAP];
    __shared__ short s_kx[VCAP];
    __shared__ int s_wsum[FIN_THREADS / 64];
    __shared__ int s_flag[3];
    __shared__ int s_head[HASH ? FIN_BUCKETS : 1];
    __shared__ unsigned short s_next[HASH ? ECAP : 1];
    const int bv = blockIdx.x;
    const int b = bv / NVAR;
    const int tid = threadIdx.x;
    int n = est_count[bv];
    if (cent_count[bv] > g.cent_cap || n > g.est_cap) {
        if (tid == 0) { overflow[b] = 1; vcount[bv] = 0; }
        return;
    }
    int np2 = 1;
    while (np2 < n) np2 <<= 1;
    for (int i = tid; i < np2; i += FIN_THREADS) s_key[i] = i < n ? est_keys[(size_t)bv * g.est_cap + i] : ~0ull;
    if (HASH) for (int i = tid; i < FIN_BUCKETS; i += FIN_THREADS) s_head[i] = -1;
    __syncthreads();
    // bitonic sort, one compare-exchange per thread and stage: thread t owns the pair (i, i | j) with bit j of i clear
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (np2 >> 1); t += FIN_THREADS) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), ixj = i | j;
                const unsigned long long a = s_key[i], c = s_key[ixj];
                const bool up = (i & k) == 0;
                if ((a > c) == up) { s_key[i] = c; s_key[ixj] = a; }
            }
            __syncthreads();
        }
    // RemoveOverlaps: circle i is kept iff no KEPT circle j < i lies within min_dist.  Resolved in parallel rounds instead
    // of the sequential sweep: a candidate whose earlier neighbours (within min_dist) are all decided is decided itself;
    // clusters around one stone settle in 2-3 rounds.  The outcome is the sequential greedy's, by induction on i.
    unsigned char* s_st = reinterpret_cast<unsigned char*>(s_kx);        // 0 undecided, 1 kept, 2 rejected (n <= ECAP bytes)
    static_assert(sizeof(short) * VCAP >= ECAP, "status bytes alias s_kx");
    for (int i = tid; i < n; i += FIN_THREADS) s_st[i] = 0;
    if (tid < 3) s_flag[tid] = 0;
    // cells of ceil(min_dist) pixels: a neighbour within min_dist lies in the 3 x 3 cells around the candidate's
    const int cs = imax(1, (int)ceilf(min_dist));
    if (HASH) {
        for (int i = tid; i < n; i += FIN_THREADS) {
            const unsigned long long key = s_key[i];
            const int cx = (int)((key >> 16) & 0xffffu) / cs, cy = (int)(key & 0xffffu) / cs;
            s_next[i] = (unsigned short)atomicExch(&s_head[(cy * 131 + cx) & (FIN_BUCKETS - 1)], i);     // -1 -> 0xffff ends a chain
        }
    }
    __syncthreads();
    const float md2 = min_dist * min_dist;
    for (int round = 0; round < n; round++) {
        bool undecided_left = false;
        for (int i = tid; i < n; i += FIN_THREADS) {
            if (s_st[i] != 0) continue;
            const unsigned long long key = s_key[i];
            const int x = (int)((key >> 16) & 0xffffu), y = (int)(key & 0xffffu);
            int verdict = 1;                                             // kept unless an earlier neighbour objects
            auto look = [&](int j) {
                const unsigned long long kj = s_key[j];
                const float ddx = (float)(x - (int)((kj >> 16) & 0xffffu)), ddy = (float)(y - (int)(kj & 0xffffu));
                if (ddx * ddx + ddy * ddy < md2) {
                    const int sj = s_st[j];
                    if (sj == 1) verdict = 2;
                    else if (sj == 0 && verdict == 1) verdict = 0;       // must wait for j
                }
            };
            if (HASH) {
                const int cx = x / cs, cy = y / cs;
                for (int oy = -1; oy <= 1 && verdict != 2; oy++)
                    for (int ox = -1; ox <= 1 && verdict != 2; ox++)
                        for (int j = s_head[((cy + oy) * 131 + cx + ox) & (FIN_BUCKETS - 1)]; j != -1 && j != 0xffff && verdict != 2; j = s_next[j])
                            if (j < i) look(j);
            } else {
                for (int j = 0; j < i && verdict != 2; j++) look(j);
            }
            if (verdict) s_st[i] = (unsigned char)verdict; else undecided_left = true;
        }
        // "somebody is still undecided" flag of this round: three slots in rotation, so that the slot of the next round can be
        // cleared here without racing with threads that have not yet read the previous round's (one barrier per round)
        if (undecided_left) s_flag[round % 3] = 1;
        if (tid == 0) s_flag[(round + 1) % 3] = 0;
        __syncthreads();
        if (s_flag[round % 3] == 0) break;
    }
    // ordered compaction of the kept circles: positions from the wave's ballot, wave totals summed through LDS
    int base = 0;
    const int lane = tid & 63, wave = tid >> 6;
    for (int c0 = 0; c0 < n; c0 += FIN_THREADS) {
        const int i = c0 + tid;
        const int keep = (i < n && s_st[i] == 1) ? 1 : 0;
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wsum[wave] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int q = 0; q < FIN_THREADS / 64; q++) { const int c = s_wsum[q]; total += c; if (q < wave) before += c; }
        const int pos = base + before + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < g.vcirc_cap) {
            const unsigned long long key = s_key[i];
            const int x = (int)((key >> 16) & 0xffffu), y = (int)(key & 0xffffu);
            const int sr = 1023 - (int)((key >> 32) & 0x3ffu);
            float* o = vcirc + ((size_t)bv * g.vcirc_cap + pos) * 3;
            o[0] = ((float)x + 0.5f) * 1.0f;
            o[1] = ((float)y + 0.5f) * 1.0f;
            o[2] = (float)sr / 2.f / 10.f * 1.0f + (float)min_r;
        }
        base += total;
        __syncthreads();
    }
    if (tid == 0) {
        const bool over = base > g.vcirc_cap;
        vcount[bv] = over ? 0 : base;
        if (over) overflow[b] = 1;
    }
}

}  // namespace i2s
