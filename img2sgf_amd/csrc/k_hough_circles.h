// cv.HoughCircles(img, HOUGH_GRADIENT, dp=1, minDist, param1, param2, minRadius, maxRadius) (img2sgf.py:180)
// after OpenCV hough.cpp HoughCirclesGradient (>= 3.4.2 / 4.x), restructured for CDNA4:
//   k_edge_bins     : edge pixels -> 8-byte records (position + fixed-point unit gradient) binned by 32x32 cell.
//   k_vote_centres  : the 2-D accumulator never exists in HBM.  Each workgroup owns a 126x126 block of
//                     accumulator cells (+1-cell apron) of TWO inputs as a 66 KB LDS tile of 16-bit counters, streams
//                     the edge bins within reach, casts the votes with LDS atomics (64 (record, direction) rays per
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
constexpr int EB = 32;           // edge bins: EB x EB pixel cells
constexpr int EB_CAP = EB * EB;  // worst case: every pixel of a bin is an edge
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
    // one block = 4 x 1 bins (128 x 32 pixels); two 16-byte map loads per thread (rows ly and ly + 16), both in flight together.
    // Round 4 (the SQ counters of round 3's kernel: 559 vector instructions per wavefront at 100 % of the SIMD cycles -- bound by its
    // arithmetic, not by the latencies its structure suggests): the edge pixels of a thread's 32 become a 32-bit MASK in ~25
    // instructions (bit 1 of a map byte says "== 2"; the four dwords of a load are OR-ed into the four low bits of each byte), the
    // masks' population counts are prefix-summed over the block, and edge pixel i of the block is found by the thread that will do
    // its gradient: a binary search over the 128 prefix sums, then the j-th set bit of that mask.  Round 3 tested every byte of
    // every dword in every wavefront (one lane with an edge in that dword sufficed) and appended through one LDS counter.
    static_assert(EB_NLD == 2 && EBT == 128 && EB_RPI == 16, "mask layout below: two rows of 16 pixels per thread");
    __shared__ unsigned s_mask[EBT];
    __shared__ unsigned short s_pre[EBT];          // inclusive prefix sums of the masks' population counts (<= 4096)
    __shared__ int s_wtot[EBT / 64];
    __shared__ int s_n[EBB_X * EBB_Y];
    const TileId tl = tile_of_block(gx, gy);               // (a plain 3-D grid in natural order: 3.9 -> 5.4 us, profiles/r04_d_edge_bins.txt)
    const int b = tl.z / NVAR, v = tl.z % NVAR;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = tl.tx * (EBB_X * EB), y0 = tl.ty * (EBB_Y * EB);
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t off = ((size_t)v * g.nb + b) * g.slot;
    const uint8_t* plane = v == 0 ? desc[b].grey : planes + off;      // variant 0 may be the source image itself (ImgDesc::grey)
    const int ppitch = v == 0 ? desc[b].gpitch : g.pitch;
    const uint8_t* map = maps + off;
    const size_t bin0 = (size_t)(b * NVAR + v) * g.bins + (size_t)tl.ty * EBB_Y * g.bw + (size_t)tl.tx * EBB_X;
    if (tid < EBB_X * EBB_Y) s_n[tid] = 0;
    // mask bit 8 q + d + 4 r  <=>  pixel 4 d + q of the thread's 16 in row ly + 16 r is an edge
    const int ly = tid >> 3, c16 = (tid & 7) * 16;
    const int xs = x0 + c16;
    unsigned M = 0;
    {
        uint4 m16[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int y = y0 + ly + r * EB_RPI;
            m16[r] = make_uint4(0u, 0u, 0u, 0u);
            if (y < h && xs < w) m16[r] = *reinterpret_cast<const uint4*>(map + rowoff(y, g.pitch) + xs);
        }
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const unsigned t0 = m16[r].x & 0x02020202u, t1 = m16[r].y & 0x02020202u, t2 = m16[r].z & 0x02020202u, t3 = m16[r].w & 0x02020202u;
            const unsigned t3a = t3 + t3;
            const unsigned u = bitop3<0xFE>(t0 >> 1, t1, t2 + t2) | (t3a + t3a);
            M |= r ? u << 4 : u;
        }
        if (xs + 16 > w) {                                   // the image ends inside these 16 pixels
            unsigned keep = 0;
            for (int p = 0; p < 16; p++) if (xs + p < w) keep |= 0x11u << (8 * (p & 3) + (p >> 2));
            M &= keep;
        }
    }
    {
        int pre = __popc(M);
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(pre, (unsigned)o); if (lane >= o) pre += t; }
        if (lane == 63) s_wtot[wave] = pre;
        __syncthreads();
        if (wave == 1) pre += s_wtot[0];
        s_pre[tid] = (unsigned short)pre;
        s_mask[tid] = M;
    }
    __syncthreads();
    const int nl = s_pre[EBT - 1];
    for (int i = tid; i < nl; i += EBT) {
        int lo = 0;
#pragma unroll
        for (int s = EBT / 2; s >= 1; s >>= 1) if ((int)s_pre[lo + s - 1] <= i) lo += s;
        int j = i - (lo ? (int)s_pre[lo - 1] : 0);
        unsigned m = s_mask[lo];
        int pos = 0, c;
        c = __popc(m & 0xffffu); if (j >= c) { j -= c; m >>= 16; pos = 16; }
        c = __popc(m & 0xffu);   if (j >= c) { j -= c; m >>= 8; pos += 8; }
        c = __popc(m & 0xfu);    if (j >= c) { j -= c; m >>= 4; pos += 4; }
        c = __popc(m & 0x3u);    if (j >= c) { j -= c; m >>= 2; pos += 2; }
        if (j >= (int)(m & 1u)) pos += 1;
        const int lx = (lo & 7) * 16 + 4 * (pos & 3) + (pos >> 3), lyy = (lo >> 3) + EB_RPI * ((pos >> 2) & 1);
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

// Votes of up to 64 (edge record, direction) items, one per lane (item = index into `bin_ent` | direction << 31, where `bin_ent` is the
// record array of the workgroup's PAIR of HoughCircles inputs: 2 x bins x EB_CAP <= 2^29 records for the largest context, so the index
// never reaches the direction bit.  Until the end of round 4 the index counted from the start of the whole context's array and passed
// 2^31 from image 256 of a pass of 1024 x 1024 images on -- wrong boards for the images behind it, found by a pass of 320).
// The lane walks its item through r = min_r .. max_r: cell = ((x * 1024 + r * sx) >> 10, (y * 1024 + r * sy) >> 10), relative to
// the first valid cell of the tile, with (sx, sy) negated for the second direction.  Cells outside [0, vx_n) x [0, vy_n)
// (outside the image or the tile) are skipped, which equals OpenCV's "break at the first cell outside the image" because a
// ray leaves the convex image only once.
template <int NSTEPS>   // > 0: the number of radius steps is known at compile time (the loop is unrolled); 0: use `nsteps`
__device__ __forceinline__ void vote_walk64(unsigned item, bool active, const uint2* __restrict__ bin_ent, unsigned ent_split,
                                            int vx_lo, int vy_lo, unsigned vx_n, unsigned vy_n, int offx, int offy, int min_r,
                                            int nsteps, unsigned* __restrict__ s_acc)
{
    // x, y: 22.10 fixed point relative to the LDS tile's first cell (the valid-cell origin (vx_lo, vy_lo) sits at (offx, offy) in
    // the tile), so that one unsigned compare per axis against the limits below is the whole range test and the shifted
    // coordinates index the tile directly.  In a tile on the image's left / top border this also lets through the cells of
    // column / row -1: they land in the apron, which there is only ever read as the neighbour of a column-0 / row-0 cell, and
    // those are never centre candidates (OpenCV scans rows / columns 1 ..).
    if (active) {
        const unsigned idx = item & 0x7fffffffu;
        const uint2 e = bin_ent[idx];                      // read a moment ago by the culling pass: an L1 / L2 hit
        int sx = (int)(short)(e.y & 0xffffu), sy = (int)(short)(e.y >> 16);
        if (item >> 31) { sx = -sx; sy = -sy; }
        int x = (((int)(e.x & 0xffffu) - vx_lo + offx) << 10) + __mul24(min_r, sx);
        int y = (((int)(e.x >> 16) - vy_lo + offy) << 10) + __mul24(min_r, sy);
        // the 16-bit half of the cell's dword is the item's HoughCircles input: a constant of the walk
        const unsigned val = idx >= ent_split ? 0x10000u : 1u;
        const unsigned xl = (vx_n + (unsigned)offx) << 10, yl = (vy_n + (unsigned)offy) << 10;
        auto step = [&]() {
            if ((unsigned)x < xl && (unsigned)y < yl)
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

// grid (tiles_x * tiles_y * nb * NVAR / 2), block 1024.
// cent_list[(b * NVAR + v) * g.cent_cap + i] = x | y << 16 of an accumulator local maximum; cent_count likewise.
// dbg_acc (optional): dense int32 accumulator, cell (x,y) of (b,v) at ((b * NVAR + v) * hmax + y) * pitch + x.
//
// Votes of one edge pixel: cells ((x*1024 +- r*sx) >> 10, (y*1024 +- r*sy) >> 10), r = min_r..max_r, that lie inside the
// image (OpenCV walks r upward and breaks at the first cell outside; the walk is a straight line from inside a convex
// image, so "break" == "skip every outside cell").  The (edge, direction, r) votes are therefore independent: a wavefront
// walks 64 (edge, direction) rays at a time, one per lane (vote_walk64).  The 2-D accumulator never exists in HBM: a workgroup
// owns the same 126 x 126 cells (+1-cell apron) of TWO HoughCircles inputs (variants 2p and 2p + 1 of one image) as one LDS tile of
// 128 x 129 dwords -- low halves variant 2p, high halves variant 2p + 1 -- and tests the 4-neighbour local-maximum rule in place.
// A cell receives at most 3 votes from each of the < 3100 edge pixels within max_r <= 30 of it, so a 16-bit half never carries into
// its neighbour; the row stride is odd, so the cells of a vertical ray spread over the banks (same-address LDS atomics serialise,
// ~4 cycles per extra lane: profiles/r01_g_lds_atomic_microbench.txt).
// Round 4: rounds 1-3 kept ONE input per workgroup of 512 threads and used the two halves for rows r and r + 64 of its tile; the
// walk step then needs "row & 63" and "max(y & 0x10000, 1)" besides the address: 10 vector + 4 scalar instructions.  With the halves
// given to two inputs the increment is a constant of the item -- which input's bin list the record came from -- and the row indexes
// the tile directly: 8 + 3.  16 wavefronts share the pair (78.5 KB of LDS: two workgroups per CU, the same eight wavefronts per SIMD):
// 15.55 -> 14.77 us per diagram.  What was built on top of it, measured and taken out again -- items that stay inside the tile walked
// without the range test (6 + 0 instructions per step), per-tile ray lists written once per record by a producer kernel with the
// lists read transposed (vote kernel 10.6 us, but the producer costs 4.3 us or more in five formulations: it is bound by its own
// ~1200 vector instructions per 128 x 32-pixel block) -- is in profiles/r04_a_vote_experiments.txt.
// (Measured earlier: plain 32-bit cells halve the resident workgroups per CU and run 1.5x slower; a branch-free variant that
// lets out-of-tile lanes add 0 to clamped cells runs 1.4x slower because of same-address conflicts.)
constexpr int VPT = 1024, VPW = VPT / 64;

template <int NSTEPS>
__global__ __launch_bounds__(VPT) void k_vote_centres(const ImgDesc* __restrict__ desc, Geo g,
                                                      const uint2* __restrict__ bin_ent, const int* __restrict__ bin_cnt,
                                                      int min_r, int max_r, int acc_thr,
                                                      unsigned* __restrict__ cent_list, int* __restrict__ cent_count,
                                                      int* __restrict__ dbg_acc, int gx, int gy)
{
    __shared__ __attribute__((aligned(16))) unsigned s_acc[VL * VASTR];         // cell (cx, cy): dword cy * VASTR + cx; low half variant 2p, high half variant 2p + 1
    __shared__ int s_ticket;
    __shared__ unsigned s_ring[VPW][VRING];
    __shared__ int s_fill[VPW];
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z / (NVAR / 2), v0 = (tl.z % (NVAR / 2)) * 2;
    const int w = desc[b].w, h = desc[b].h;
    const int cx0 = tl.tx * VT, cy0 = tl.ty * VT;    // first interior cell
    if (cx0 >= w || cy0 >= h) return;
    const int tid = threadIdx.x;
    const int bv = b * NVAR + v0;
    static_assert((VL * VASTR) % 4 == 0, "the tile is zeroed in 16-byte stores");
    for (int i = tid; i < VL * VASTR / 4; i += VPT) reinterpret_cast<uint4*>(s_acc)[i] = make_uint4(0u, 0u, 0u, 0u);
    if (tid == 0) s_ticket = VPW;                  // bins 0 .. 15 are taken by the waves' first round
    __syncthreads();
    // LDS tile covers cells [lx0, lx0 + VL) x [ly0, ly0 + VL); edge pixels within max_r of it can vote into it
    const int lx0 = cx0 - 1, ly0 = cy0 - 1;
    const int bx0 = imax(lx0 - max_r, 0) / EB, bx1 = imin(lx0 + VL - 1 + max_r, w - 1) / EB;
    const int by0 = imax(ly0 - max_r, 0) / EB, by1 = imin(ly0 + VL - 1 + max_r, h - 1) / EB;
    const int nbx = bx1 - bx0 + 1, nbin = nbx * (by1 - by0 + 1);      // per variant, <= 49; bins nbin .. 2 nbin - 1 are variant 2p + 1's
    // cells of this tile that exist in the image: one unsigned compare per axis covers "inside image" and "inside tile"
    const int vx_lo = imax(lx0, 0), vy_lo = imax(ly0, 0);
    const unsigned vx_n = (unsigned)(imin(lx0 + VL, w) - vx_lo), vy_n = (unsigned)(imin(ly0 + VL, h) - vy_lo);
    const int offx = vx_lo - lx0, offy = vy_lo - ly0;      // valid-cell origin inside the LDS tile (0 or 1)
    const int nsteps = max_r - min_r + 1;          // <= 31
    const int lane = tid & 63, wave = tid >> 6;
    bin_ent += (size_t)bv * g.bins * EB_CAP;          // records of this pair of inputs; bin_of / items / the walk index from here
    const unsigned ent_split = (unsigned)g.bins * (unsigned)EB_CAP;                // first record index of variant 2p + 1
    const int xl = (int)((vx_n + (unsigned)offx) << 10), yl = (int)((vy_n + (unsigned)offy) << 10);
    // one wavefront per bin: a coalesced 512-byte load brings 64 edge records, every lane tests whether ITS record's rays can touch
    // the tile at all, the surviving (record, direction) items are compacted into the wavefront's LDS ring, and whenever 64 are
    // waiting they are walked together.  The reach window spans at most 7 x 7 bins per variant; lane q holds bin q's index and count
    // (the division by the window width is the expensive part), the walk fetches them with v_readlane.  Bins are handed out
    // dynamically (LDS ticket) because their populations differ a lot (grid lines concentrate in a few bins): with a static split
    // the waves of a workgroup spent 40 % of their time waiting for the slowest one at the barrier.  Each wave keeps one bin in
    // flight ahead of the one it culls.
    int my_cnt_a = 0, my_cnt_b = 0, my_bin = 0;
    if (lane < nbin) {
        my_bin = (by0 + lane / nbx) * g.bw + (bx0 + lane % nbx);                // relative to the pair's first bin
        my_cnt_a = bin_cnt[(size_t)bv * g.bins + my_bin];
        my_cnt_b = bin_cnt[(size_t)(bv + 1) * g.bins + my_bin];
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
    int fill = 0;
    const unsigned long long below = (1ull << lane) - 1ull;
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
            // The reach test is made per DIRECTION (a ray that points away from the tile would only occupy a lane for 30 steps).  On
            // each axis the ray's coordinates run monotonically from X0 + d min_r s to X0 + d max_r s, d = +-1: it can touch the tile
            // only if that span meets [0, limit) on both axes -- a superset, the per-step test decides.
            // (24-bit multiplies: |s| <= 1024 and the radius is far below 2^23; v_mul_lo_u32 is quarter rate)
            bool in_p = false, in_n = false;
            if (k0 + lane < n_cur) {
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
                vote_walk64<NSTEPS>(ring[lane], true, bin_ent, ent_split, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
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
    // Every wavefront ends with fewer than 64 items in its ring, and a walk costs 30 steps whatever its fill: the sixteen remainders
    // are walked as ONE list, 64 items per wavefront.  Lane l of wavefront k takes item 64 k + l of the concatenated rings.
    if (lane == 0) s_fill[wave] = fill;
    __syncthreads();
    {
        int before = 0, r = 0, tot = 0;
        const int gi = wave * 64 + lane;
#pragma unroll
        for (int k = 0; k < VPW; k++) {
            const int f = s_fill[k];
            if (gi >= tot + f) { before = tot + f; r = k + 1; }
            tot += f;
        }
        const int cnt = imin(imax(tot - wave * 64, 0), 64);
        if (cnt > 0) {
            unsigned it = 0;
            if (lane < cnt) it = s_ring[r][gi - before];
            vote_walk64<NSTEPS>(it, lane < cnt, bin_ent, ent_split, vx_lo, vy_lo, vx_n, vy_n, offx, offy, min_r, nsteps, s_acc);
        }
    }
    __syncthreads();
    // centre candidates: cells (x,y), 1 <= x <= w-1, 1 <= y <= h-1 (OpenCV scans padded rows 1..H, cols 1..W of an accumulator whose
    // votes sit at unpadded indices; cells x == W or y == H hold no votes): interior cells (tx, ty), 1 <= tx, ty <= VT, of either half
#define I2S_CELL(cx, cy, hh) ((int)((s_acc[(cy) * VASTR + (cx)] >> ((hh) * 16)) & 0xffffu))
    if (dbg_acc) {
        for (int i = tid; i < VT * VT; i += VPT) {
            const int ty = i / VT + 1, tx = i - (ty - 1) * VT + 1;
            const int x = lx0 + tx, y = ly0 + ty;
            if (x < w && y < h) {
                dbg_acc[((size_t)bv * g.hmax + y) * g.pitch + x] = I2S_CELL(tx, ty, 0);
                dbg_acc[((size_t)(bv + 1) * g.hmax + y) * g.pitch + x] = I2S_CELL(tx, ty, 1);
            }
        }
    }
    // almost all cells hold fewer votes than the threshold in BOTH halves: four cells of a row are rejected together -- per dword
    // (v + K) & 0x80008000 with K = 0x7fff - threshold in both halves is non-zero iff a half exceeds the threshold (a half is at
    // most 3 x 3100 votes: no carry between the halves) -- in 4 trips of the workgroup instead of 16 with a division by VT each
    {
        static_assert(VT == 126 && VL == 128, "32 groups of four cells per interior row; the last group holds two cells, the apron and nothing");
        const unsigned K = (unsigned)(0x7fff - iclamp(acc_thr, 0, 0x7fff)) * 0x00010001u;
        for (int i = tid; i < VT * 32; i += VPT) {
            const int ty = (i >> 5) + 1, tx0 = 1 + 4 * (i & 31);
            const unsigned* row = s_acc + ty * VASTR + tx0;
            const unsigned c0 = row[0], c1 = row[1], c2 = tx0 + 2 <= VT ? row[2] : 0u, c3 = tx0 + 3 <= VT ? row[3] : 0u;
            if ((bitop3<0xFE>(c0 + K, c1 + K, c2 + K) | (c3 + K)) & 0x80008000u) {
                const unsigned cv[4] = {c0, c1, c2, c3};
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const unsigned v2 = cv[q];
                    if (!((v2 + K) & 0x80008000u)) continue;
                    const int tx = tx0 + q;
                    const int x = lx0 + tx, y = ly0 + ty;
                    if (x >= w || x < 1 || y >= h || y < 1) continue;
#pragma unroll
                    for (int hh = 0; hh < 2; hh++) {
                        const int a = hh ? (int)(v2 >> 16) : (int)(v2 & 0xffffu);
                        if (a <= acc_thr) continue;
                        if (a > I2S_CELL(tx - 1, ty, hh) && a >= I2S_CELL(tx + 1, ty, hh) && a > I2S_CELL(tx, ty - 1, hh) && a >= I2S_CELL(tx, ty + 1, hh)) {
                            const int k = atomicAdd(&cent_count[bv + hh], 1);
                            if (k < g.cent_cap) cent_list[(size_t)(bv + hh) * g.cent_cap + k] = (unsigned)x | ((unsigned)y << 16);
                        }
                    }
                }
            }
        }
    }
#undef I2S_CELL
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
