// Canny (img2sgf.py:162 and the Canny inside every cv.HoughCircles call, :180) as integer HIP kernels.
// Follows OpenCV canny.cpp: Sobel 3x3 (CV_16S, BORDER_REPLICATE), L1 magnitude, per-pixel max-magnitude
// channel for colour input, non-maximum suppression with the TG22 fixed-point sectors, strict '>'
// thresholds, 8-connected hysteresis.  Map values: 0 = weak candidate, 1 = no edge, 2 = edge.
#pragma once
#include "i2s_types.h"
#include "tile_io.h"
#include <gfx950_ops.h>

namespace i2s {

constexpr int CT_W = 64;   // NMS output tile
constexpr int CT_H = 32;
// hysteresis works on the same 64 x 32 tiles.  Tiles that hold weak pixels are appended to a worklist by the NMS kernels:
// wl[0] = count, wl[1 + i] = (m * nb + b) * g.tiles + ty * g.tw + tx; only those tiles are ever visited again.


// One hysteresis pass over the tiles of a worklist (all maps of the phase).  A fixed, small grid of workgroups strides over
// the list, one WAVEFRONT per listed 64x32 tile: lane r holds row y0 - 1 + r of the tile (rows -1 and 32 are the read-only
// apron) as two 64-bit masks, S (edge) and W (weak candidate).  One step of "a weak pixel becomes an edge iff an 8-neighbour
// is an edge" is then a handful of 64-bit operations for the whole tile: neighbours in the row are shifts, rows above / below
// come from the neighbouring lanes, and a row is flooded along its weak runs at once by a carry chain
// (((M + S) ^ M) & M, M = W | S, in both bit orders).  The tile reaches its local fixed point in as many steps as its
// longest chain spans ROWS, with no workgroup barrier and no LDS -- scans of photographs, whose contours wander across
// whole tiles, spent a third of their GPU time in the previous byte-per-thread version (up to 300 us per launch).
// The host launches passes back to back.  Pass 0 walks the worklist of the Sobel / NMS kernel (tiles that hold weak pixels); a tile
// leaves its visit at its local fixed point, so it has to be looked at again only if its APRON changes -- a tile that promotes
// pixels on a border queues the neighbours that border faces (the bottom row: the tile below; the bottom row AND the right
// column, i.e. the corner pixel: also the one below right ...) for the next pass: queue[(pass + 1) & 1], flags[pass] = its
// length, marks[tile] = stamp of the latest pass the tile is queued for (atomicMax: a tile enters a queue once).  Pass p >= 1
// walks queue[p & 1] only, and returns at once when it is empty (flags[p - 1] == 0): the fixed point.  The result is the unique
// fixed point, independent of scheduling and of the list order.  (Round 3's first version kept a per-tile "borders changed"
// record that every pass looked up for all nine neighbours of every worklist tile: on noisy scans, where every tile is on the
// worklist, that scan alone was 25-35 us per pass and 256 images.)
// maps points at map 0; map m of image b at (m * nb + b) * slot.  `edges` (non-null for the main Canny's phase, whose worklist
// holds map-0 tiles only) receives 255 / 0 for every rewritten dword: together with the NMS kernel's output that is the edge
// image of img2sgf.py:162.  grid (HY_BLOCKS), block 256 = 4 independent wavefronts.
constexpr int HY_BLOCKS = 2048;

// flag bytes (0 / 1): byte i of v == 2, byte i of v == 0 (map bytes are 0, 1 or 2)
__device__ __forceinline__ unsigned hy_flag_edge(unsigned v) { return (v >> 1) & ~v & 0x01010101u; }
__device__ __forceinline__ unsigned hy_flag_weak(unsigned v) { return ~(v | (v >> 1)) & 0x01010101u; }
// 8 mask bits from the flag bytes of two dwords: a byte-wise dot product with the bit weights (v_dot4_u32_u8)
__device__ __forceinline__ unsigned hy_bits8(unsigned f_lo, unsigned f_hi)
{
    return __builtin_amdgcn_udot4(f_hi, 0x80402010u, __builtin_amdgcn_udot4(f_lo, 0x08040201u, 0u, false), false);
}

// flood the seeds S along the runs of M (S subset of M) towards higher bits
__device__ __forceinline__ unsigned long long hy_fill_up(unsigned long long S, unsigned long long M) { return (((M + S) ^ M) & M) | S; }

// one pass over this workgroup's share of the worklist (entries first, first + stride, ...; one wavefront per entry)
__device__ __forceinline__ void hysteresis_pass(const ImgDesc* __restrict__ desc, const Geo& g, uint8_t* __restrict__ maps,
                                                uint8_t* __restrict__ edges, int* __restrict__ flags, int pass,
                                                const int* __restrict__ wl, int* __restrict__ queues, size_t queue_half,
                                                int* __restrict__ marks, int stamp_base)
{
    const int* src = pass == 0 ? wl + 1 : queues + (size_t)(pass & 1) * queue_half;
    int* dst = queues + (size_t)((pass + 1) & 1) * queue_half;
    const int nwl = pass == 0 ? wl[0] : load_agent(&flags[pass - 1]);
    const int stamp = stamp_base + pass + 1;                              // "queued for pass + 1"
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int e = blockIdx.x * 4 + wave; e < nwl; e += gridDim.x * 4) {
        const int key = load_agent(&src[e]);
        const int mb = key / g.tiles, tile = key - mb * g.tiles;       // mb = m * nb + b
        const int b = mb % g.nb;
        const int ty_ = tile / g.tw, tx_ = tile - ty_ * g.tw;
        const int w = desc[b].w, h = desc[b].h;
        const int x0 = tx_ * CT_W, y0 = ty_ * CT_H;
        const size_t tbase = (size_t)mb * g.tiles;
        uint8_t* mp = maps + (size_t)mb * g.slot;
        const int y = y0 - 1 + lane;
        const bool row_in = lane < CT_H + 2 && y >= 0 && y < h;
        const bool core = row_in && lane >= 1 && lane <= CT_H;
        uint4 q[4];
        unsigned left = 0x01010101u, right = 0x01010101u;
#pragma unroll
        for (int k = 0; k < 4; k++) q[k] = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u);
        if (row_in) {
            const uint8_t* row = mp + rowoff(y, g.pitch) + x0;
#pragma unroll
            for (int k = 0; k < 4; k++) q[k] = *reinterpret_cast<const uint4*>(row + 16 * k);
            if (x0 > 0) left = *reinterpret_cast<const unsigned*>(row - 4);
            if (x0 + CT_W < w) right = *reinterpret_cast<const unsigned*>(row + CT_W);
        }
        unsigned s_lo = 0, s_hi = 0, w_lo = 0, w_hi = 0;                   // 32 columns each
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const unsigned e01 = hy_bits8(hy_flag_edge(q[k].x), hy_flag_edge(q[k].y)), e23 = hy_bits8(hy_flag_edge(q[k].z), hy_flag_edge(q[k].w));
            const unsigned k01 = hy_bits8(hy_flag_weak(q[k].x), hy_flag_weak(q[k].y)), k23 = hy_bits8(hy_flag_weak(q[k].z), hy_flag_weak(q[k].w));
            const unsigned e16 = e01 | (e23 << 8), k16 = k01 | (k23 << 8);   // columns 16 k .. 16 k + 15
            if (k < 2) { s_lo |= e16 << (16 * k); w_lo |= k16 << (16 * k); }
            else { s_hi |= e16 << (16 * (k - 2)); w_hi |= k16 << (16 * (k - 2)); }
        }
        unsigned long long S = (unsigned long long)s_lo | ((unsigned long long)s_hi << 32);
        unsigned long long W = (unsigned long long)w_lo | ((unsigned long long)w_hi << 32);
        const int ncol = imin(w - x0, CT_W);                               // columns of the tile inside the image
        const unsigned long long colmask = ncol >= 64 ? ~0ull : ((1ull << ncol) - 1ull);
        S &= colmask;
        W = core ? (W & colmask) : 0ull;                                   // apron rows and rows outside the image never change
        const bool aL = (left >> 24) == 2u, aR = (right & 0xffu) == 2u;    // edge flags of columns x0 - 1 and x0 + 64
        const unsigned long long S0 = S;
        const unsigned long long M = W | S;
        const unsigned long long rM = __brevll(M);
        const int up = lane > 0 ? lane - 1 : 0, dn = lane < 63 ? lane + 1 : 63;
        // side columns: a pixel of column 0 (63) also neighbours column -1 (64) of its own and the adjacent rows
        const int sideL = (aL ? 1 : 0) | (__shfl(aL ? 1 : 0, up)) | (__shfl(aL ? 1 : 0, dn));
        const int sideR = (aR ? 1 : 0) | (__shfl(aR ? 1 : 0, up)) | (__shfl(aR ? 1 : 0, dn));
        const unsigned long long side = (sideL ? 1ull : 0ull) | (sideR ? (1ull << 63) : 0ull);
        for (;;) {
            // along the row, both directions, to completion (up from the lowest seed of a run, then down from its top)
            unsigned long long T = hy_fill_up(S, M);
            T |= __brevll(hy_fill_up(__brevll(T), rM));
            // one step to the rows above and below (and their diagonals), and from the side columns
            const unsigned long long a = __shfl(T, up), c = __shfl(T, dn);
            unsigned long long N = (lane > 0 ? a : 0ull) | (lane < 63 ? c : 0ull);
            N = N | (N << 1) | (N >> 1) | side;
            T |= W & N;
            const bool changed = T != S;
            S = T;
            if (__ballot(changed) == 0ull) break;
        }
        const unsigned long long promoted = core ? (S & ~S0) : 0ull;
        const unsigned long long rows_changed = __ballot(promoted != 0ull);      // bit r = lane r = tile row r - 1
        if (rows_changed == 0ull) continue;
        const int borders = (int)((rows_changed >> 1) & 1ull) | ((int)((rows_changed >> CT_H) & 1ull) << 1) |
                            (__ballot((promoted & 1ull) != 0ull) ? 4 : 0) | (__ballot((promoted >> 63) != 0ull) ? 8 : 0);
        if (promoted) {
            uint8_t* row = mp + rowoff(y, g.pitch) + x0;
            uint8_t* erow = edges ? edges + (size_t)mb * g.slot + rowoff(y, g.pitch) + x0 : nullptr;      // mb == b in the main phase
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const unsigned d[4] = {q[k].x, q[k].y, q[k].z, q[k].w};
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const unsigned nib = (unsigned)(promoted >> (16 * k + 4 * i)) & 0xfu;
                    if (nib) {
                        // promoted pixels were weak (byte 0): OR-ing 2 in makes them edges; dwords are written whole (columns
                        // beyond the image inside the pitch padding keep what they held)
                        const unsigned v = d[i] | (((nib * 0x00204081u) & 0x01010101u) << 1);
                        *reinterpret_cast<unsigned*>(row + 16 * k + 4 * i) = v;
                        if (erow) *reinterpret_cast<unsigned*>(erow + 16 * k + 4 * i) = ((v >> 1) & ~v & 0x01010101u) * 0xffu;
                    }
                }
            }
        }
        // lane k queues neighbour k (k = 4 is the tile itself) if the borders that face it changed: 1 top row, 2 bottom row, 4 left
        // column, 8 right column
        if (lane < 9 && lane != 4 && borders != 0) {
            const int ntx = (w + CT_W - 1) / CT_W, nty = (h + CT_H - 1) / CT_H;
            const int ddx = lane % 3 - 1, ddy = lane / 3 - 1;
            const int tx = tx_ + ddx, ty = ty_ + ddy;
            const int need = (ddy > 0 ? 2 : (ddy < 0 ? 1 : 0)) | (ddx > 0 ? 8 : (ddx < 0 ? 4 : 0));
            if (tx >= 0 && tx < ntx && ty >= 0 && ty < nty && (borders & need) == need) {
                const int nkey = (int)tbase + ty * g.tw + tx;
                if (atomicMax(&marks[nkey], stamp) < stamp) dst[atomicAdd(&flags[pass], 1)] = nkey;
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_hysteresis(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ maps,
                                                    uint8_t* __restrict__ edges, int* __restrict__ flags, int pass,
                                                    const int* __restrict__ wl, int* __restrict__ queues, size_t queue_half,
                                                    int* __restrict__ marks, int stamp_base)
{
    static_assert(CT_W == 64 && CT_H + 2 <= 64, "one 64-bit mask per row, one lane per row incl. the apron");
    if (pass > 0 && flags[pass - 1] == 0) return;
    hysteresis_pass(desc, g, maps, edges, flags, pass, wl, queues, queue_half, marks, stamp_base);
}

// The tail of a phase: ONE launch behind the `first_pass` plain launches (a number the host adapts to what the previous calls
// needed), which runs whatever passes are still necessary inside the kernel -- a grid of HY_TAIL_BLOCKS workgroups with a grid-wide
// barrier between passes -- and reports how many passes the phase took: info[0] = passes that ran until one queued nothing for the next
// (first_pass if the plain launches had already converged), or -1 when the budget `max_pass` or a barrier timeout was hit (the
// host then redoes the device pass with plain launches).  On diagrams the tail finds flags[first_pass - 1] == 0 and returns at
// once: a phase costs first_pass + 1 launches instead of a fixed budget of six.
__global__ __launch_bounds__(256) void k_hysteresis_tail(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ maps,
                                                         uint8_t* __restrict__ edges, int* __restrict__ flags, int first_pass, int max_pass,
                                                         const int* __restrict__ wl, int* __restrict__ queues, size_t queue_half,
                                                         int* __restrict__ marks, int stamp_base,
                                                         int* __restrict__ counter, int* __restrict__ info)
{
    __shared__ int s_ok;
    int target = 0;
    int pass = first_pass;
    bool ok = true;
    for (; pass < max_pass; pass++) {
        if (load_agent(&flags[pass - 1]) == 0) break;                    // the previous pass queued no tile: fixed point
        hysteresis_pass(desc, g, maps, edges, flags, pass, wl, queues, queue_half, marks, stamp_base);
        ok = grid_barrier(counter, target, &s_ok);
        if (!ok) break;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        int used = ok ? -1 : -2;                                           // -1: the budget is spent, -2: a grid barrier timed out
        if (ok) for (int q = 0; q < pass; q++) if (load_agent(&flags[q]) == 0) { used = q + 1; break; }
        info[0] = used;
    }
}

}  // namespace i2s
