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
#ifndef NMS_FW
#define NMS_FW 16   // strips per row of a wavefront's NMS footprint (16 = whole strip rows: measured equal to 16 x 16 pixel blocks, stores coalesce better)
#endif
constexpr int NMS_TPB = 4;  // consecutive tiles (along x) handled by one workgroup of k_sobel_nms_planes, software-pipelined
// hysteresis works on the same 64 x 32 tiles.  Tiles that hold weak pixels are appended to a worklist by the NMS kernels:
// wl[0] = count, wl[1 + i] = (m * nb + b) * g.tiles + ty * g.tw + tx; only those tiles are ever visited again.

// Sobel + NMS of one tile.  src(y,x,c) = sp[y*sstride + x*CN + c].
template <int CN>
__device__ __forceinline__ void sobel_nms_tile(const uint8_t* __restrict__ sp, int sstride, int w, int h,
                                               int low, int high, uint8_t* __restrict__ mp, uint8_t* __restrict__ ep, int mpitch,
                                               int* __restrict__ weak_wl, int weak_key, int tile_x, int tile_y)
{
    __shared__ int s_weak;
    constexpr int SW = CT_W + 4, SH = CT_H + 4;   // source tile with 2-px apron
    constexpr int MW = CT_W + 2, MH = CT_H + 2;   // gradient tile with 1-px apron
    __shared__ __attribute__((aligned(4))) uint8_t s_src[SH][SW * CN + 4];
    __shared__ short s_dx[MH][MW], s_dy[MH][MW];
    __shared__ unsigned short s_mag[MH][MW];
    const int x0 = tile_x * CT_W, y0 = tile_y * CT_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    if (tid == 0) s_weak = 0;
    if ((SW * CN) % 4 == 0 && x0 - 2 >= 0 && x0 - 2 + SW <= w && y0 - 2 >= 0 && y0 - 2 + SH <= h) {
        // tile (with its apron) wholly inside the image: rows of SW * CN bytes as (unaligned) dword loads -- byte loads of
        // interleaved channels keep the texture-address unit busy for 4x as many instructions
        constexpr int RW = SW * CN / 4;
        for (int i = tid; i < SH * RW; i += 256) {
            const int ly = i / RW, c4 = i - ly * RW;
            unsigned v;
            __builtin_memcpy(&v, sp + (size_t)(y0 + ly - 2) * sstride + (size_t)(x0 - 2) * CN + 4 * c4, 4);
            *reinterpret_cast<unsigned*>(&s_src[ly][4 * c4]) = v;
        }
    } else {
        for (int i = tid; i < SH * SW; i += 256) {
            const int ly = i / SW, lx = i - ly * SW;
            const int gy = iclamp(y0 + ly - 2, 0, h - 1), gx = iclamp(x0 + lx - 2, 0, w - 1);
            const uint8_t* p = sp + (size_t)gy * sstride + (size_t)gx * CN;
#pragma unroll
            for (int c = 0; c < CN; c++) s_src[ly][lx * CN + c] = p[c];
        }
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
        ep[(size_t)gy * mpitch + gx] = out == 2 ? 255 : 0;      // the edge image (img2sgf.py:162); hysteresis adds the promoted pixels
        if (out == 0) s_weak = 1;
    }
    __syncthreads();
    if (tid == 0 && s_weak) weak_wl[1 + atomicAdd(&weak_wl[0], 1)] = weak_key;
}

// Main Canny (map 0) on the source image: grid (tiles_x, tiles_y, nb).
template <int CN>
__global__ __launch_bounds__(256) void k_sobel_nms_src(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ map0,
                                                       uint8_t* __restrict__ edges, int low, int high, int* __restrict__ weak,
                                                       int gx, int gy)
{
    const TileId t = tile_of_block(gx, gy);
    const int b = t.z;
    const ImgDesc im = desc[b];
    if (im.cn != CN) return;
    sobel_nms_tile<CN>(im.src, im.sstride, im.w, im.h, low, high, map0 + (size_t)b * g.slot, edges + (size_t)b * g.slot, g.pitch,
                       weak, (int)((size_t)b * g.tiles + (size_t)t.ty * g.tw + t.tx), t.tx, t.ty);
}

// Sobel + NMS on single-channel PLANES, 4 pixels per thread with dword LDS traffic.
//   main_mode == 0: HoughCircles' internal Canny of variants [v_first, v_first + gridDim.z / nb): plane v -> map 1 + v.
//   main_mode == 1: the main Canny (img2sgf.py:162) of greyscale sources: plane 0 (grey == source) -> map 0
//                   (colour sources go through k_sobel_nms_src<3>); also writes the edge image (255 where the map says
//                   "edge") into `edges`, which the map-0 hysteresis then completes.
//   main_mode == 2: both at once for variant 0 (the grey plane): HoughCircles' Canny (low, high) -> map 1 and, for greyscale
//                   sources, the main Canny (low, high_main) -> map 0 + edges.  Valid when both use the same low threshold
//                   (the reference's 50 == 100 / 2): the gradients, sectors and suppression decisions are then shared and
//                   only the strong / weak split differs.
// grid (tiles_x, tiles_y, nb * nvariants), block 256, tile 64 x 32 outputs.
// planes = variant plane 0 base, maps = map 0 base.
__global__ __launch_bounds__(256) void k_sobel_nms_planes(const ImgDesc* __restrict__ desc, Geo g, const uint8_t* __restrict__ planes,
                                                          uint8_t* __restrict__ maps, uint8_t* __restrict__ edges, int v_first, int low,
                                                          int high, int high_main, int main_mode, int* __restrict__ weak,
                                                          int* __restrict__ weak_main, int gx, int gy)
{
    __shared__ int s_weak, s_weak0;
    constexpr int SROWS = CT_H + 4, SWORDS = CT_W / 4 + 4, SSTR = SWORDS + 1;   // source rows y0-2.., x0-8 .. x0+72
    constexpr int MROWS = CT_H + 2, MSTRIPS = CT_W / 4 + 2, MSTR = 2 * MSTRIPS + 1;   // mag rows y0-1.., x0-4 .. x0+68 (u16 pairs)
    __shared__ unsigned s_src[SROWS * SSTR];
    __shared__ unsigned s_mag[MROWS * MSTR];
    __shared__ unsigned s_grad[CT_H * (CT_W / 4) * 4];      // dx01, dx23, dy01, dy23 of the core strips
    // gx counts GROUPS of NMS_TPB tiles; the workgroup walks its group left to right and fetches tile t+1 into registers
    // while it computes tile t (the tile kernels are otherwise latency-bound: load -> wait -> compute -> store)
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z % g.nb;
    const int v = main_mode ? 0 : v_first + tl.z / g.nb;
    if (main_mode == 1 && desc[b].cn != 1) return;
    const bool main_out = main_mode != 0 && desc[b].cn == 1;     // writes map 0 + edges
    if (main_mode == 1) high = high_main;
    const int w = desc[b].w, h = desc[b].h;
    const int y0 = tl.ty * CT_H;
    if (tl.tx * NMS_TPB * CT_W >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    // variant 0 is the grey plane, which may be the source image itself (ImgDesc::grey)
    const uint8_t* plane = v == 0 ? desc[b].grey : planes + ((size_t)v * g.nb + b) * g.slot;
    const int ppitch = v == 0 ? desc[b].gpitch : g.pitch;
    // first output: the variant's map (modes 0, 2) or map 0 (mode 1); second output (mode 2 only): map 0
    const int m_first = main_mode == 1 ? 0 : 1 + v;
    uint8_t* mp = maps + ((size_t)m_first * g.nb + b) * g.slot;
    uint8_t* mp0 = (main_mode == 2 && main_out) ? maps + (size_t)b * g.slot : nullptr;
    uint8_t* ep = main_out ? edges + (size_t)b * g.slot : nullptr;
    int* weak_first = main_mode == 1 ? weak_main : weak;
    TileRegs<SROWS, SWORDS, 256, BORDER_REPL> pre;
    pre.fetch(plane, ppitch, w, h, tl.tx * NMS_TPB * CT_W - 8, y0 - 2, tid);
    for (int tt = 0; tt < NMS_TPB; tt++) {
    const int tile_x = tl.tx * NMS_TPB + tt;
    const int x0 = tile_x * CT_W;
    if (x0 >= w) break;
    if (tid == 0) { s_weak = 0; s_weak0 = 0; }
    pre.park<SSTR>(s_src, tid);
    __syncthreads();
    if (tt + 1 < NMS_TPB && x0 + CT_W < w) pre.fetch(plane, ppitch, w, h, x0 + CT_W - 8, y0 - 2, tid);
    // gradient strips: strip (ry, s) covers pixels x = x0 - 4 + 4s .. +3 of image row y0 - 1 + ry.
    // Two pixels per register (16-bit lanes, v_pk_* instructions): column sums / row differences of the 3x6
    // neighbourhood, then dx = col[+1] - col[-1], dy = dif[-1] + 2 dif[0] + dif[+1], mag = |dx| + |dy|.
    constexpr int NSTRIPS = MROWS * MSTRIPS;          // 34 * 18 = 612 gradient strips (core + apron)
    constexpr int PER = (NSTRIPS + 255) / 256;        // 3
    constexpr int NCORE = CT_H * (CT_W / 4);          // 512 core strips: exactly 2 per thread in the NMS phase
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int i = tid + k * 256;
        if (i < NSTRIPS) {
            const int ry = i / MSTRIPS, s = i - ry * MSTRIPS;
            const unsigned* p0 = s_src + ry * SSTR + s;
            v2s ra[3], rb[3], rc[3];          // pixel pairs (-1,0), (1,2), (3,4) of the three rows
#pragma unroll
            for (int j = 0; j < 3; j++) {
                const unsigned a = p0[j * SSTR], b0 = p0[j * SSTR + 1], c = p0[j * SSTR + 2];
                ra[j] = pk_from(__builtin_amdgcn_perm(b0, a, 0x0c040c03u));
                rb[j] = pk_from(__builtin_amdgcn_perm(b0, b0, 0x0c020c01u));
                rc[j] = pk_from(__builtin_amdgcn_perm(c, b0, 0x0c040c03u));
            }
            const v2s ca = ra[0] + ra[1] + ra[1] + ra[2], cb = rb[0] + rb[1] + rb[1] + rb[2], cc = rc[0] + rc[1] + rc[1] + rc[2];
            const v2s da = ra[2] - ra[0], db = rb[2] - rb[0], dc = rc[2] - rc[0];
            v2s dx01 = cb - ca, dx23 = cc - cb;
            const v2s m01 = pk_from(__builtin_amdgcn_alignbit(pk_bits(db), pk_bits(da), 16));    // (dif0, dif1)
            const v2s m23 = pk_from(__builtin_amdgcn_alignbit(pk_bits(dc), pk_bits(db), 16));    // (dif2, dif3)
            v2s dy01 = da + m01 + m01 + db, dy23 = db + m23 + m23 + dc;
            const int gy = y0 - 1 + ry, gx0 = x0 - 4 + 4 * s;
            if (gy < 0 || gy >= h || gx0 < 0 || gx0 + 3 >= w) {
                // strip touches the image border: gradients (hence magnitudes) outside the image are 0
                const bool row_ok = gy >= 0 && gy < h;
                unsigned k01 = 0, k23 = 0;
                if (row_ok && gx0 >= 0 && gx0 < w) k01 |= 0x0000ffffu;
                if (row_ok && gx0 + 1 >= 0 && gx0 + 1 < w) k01 |= 0xffff0000u;
                if (row_ok && gx0 + 2 >= 0 && gx0 + 2 < w) k23 |= 0x0000ffffu;
                if (row_ok && gx0 + 3 >= 0 && gx0 + 3 < w) k23 |= 0xffff0000u;
                dx01 = pk_from(pk_bits(dx01) & k01); dy01 = pk_from(pk_bits(dy01) & k01);
                dx23 = pk_from(pk_bits(dx23) & k23); dy23 = pk_from(pk_bits(dy23) & k23);
            }
            const v2s mg01 = pk_abs(dx01) + pk_abs(dy01), mg23 = pk_abs(dx23) + pk_abs(dy23);
            s_mag[ry * MSTR + 2 * s] = pk_bits(mg01);
            s_mag[ry * MSTR + 2 * s + 1] = pk_bits(mg23);
            if (ry >= 1 && ry <= CT_H && s >= 1 && s <= CT_W / 4) {
                // core strip: park the gradient for the NMS phase, which is mapped densely onto the 512 core strips
                unsigned* pg = s_grad + ((ry - 1) * (CT_W / 4) + (s - 1)) * 4;
                pg[0] = pk_bits(dx01); pg[1] = pk_bits(dx23); pg[2] = pk_bits(dy01); pg[3] = pk_bits(dy23);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NCORE / 256; k++) {
        // A wavefront's 64 strips form a compact NMS_FW x (64 / NMS_FW) block of strips (4 NMS_FW x 64 / NMS_FW pixels) rather
        // than whole strip rows: the suppression below is skipped per strip, but a wavefront only saves the time when ALL its
        // strips skip, and on line art a compact block lies between the lines far more often than a 64-pixel-wide band does.
        constexpr int NMS_FH = 64 / NMS_FW, NMS_BX = (CT_W / 4) / NMS_FW;
        const int fb = (tid >> 6) + 4 * k, fl = tid & 63;                 // footprint index / strip inside it
        const int ry = (fb / NMS_BX) * NMS_FH + fl / NMS_FW + 1, s = (fb % NMS_BX) * NMS_FW + fl % NMS_FW + 1;
        const int ci = (ry - 1) * (CT_W / 4) + (s - 1);
        const int gy = y0 - 1 + ry, gx0 = x0 - 4 + 4 * s;
        if (gy >= h || gx0 >= w) continue;
        unsigned outw = 0x01010101u, outw0 = 0x01010101u;
        const unsigned mb01 = s_mag[ry * MSTR + 2 * s], mb23 = s_mag[ry * MSTR + 2 * s + 1];
        const int mxall = imax(imax((int)(mb01 & 0xffffu), (int)(mb01 >> 16)), imax((int)(mb23 & 0xffffu), (int)(mb23 >> 16)));
        if (mxall > low) {
            // Non-maximum suppression of the strip's 4 pixels as two pairs in 16-bit lanes (magnitudes <= 2040, gradients
            // <= 1020): comparisons are packed subtractions whose sign is spread over the lane; the sector tests
            // |dy| * 2^15 < |dx| * 13573 and |dy| * 2^15 > |dx| * 79109 become |dy| <= q and |dy| > 2 |dx| + q with
            // q = floor(|dx| * 13573 / 2^15) = (|dx| * 53 + (|dx| * 5 >> 8)) >> 7 (13573 = 53 * 256 + 5 is odd, so the quotient
            // is never exact for |dx| > 0, and for |dx| = |dy| = 0 the magnitude is 0 and nothing is kept anyway).
            const unsigned* pg = s_grad + ci * 4;
            unsigned L[3][2], C[3][2], R[3][2];       // per row (above, this, below) and pair: left / centre / right magnitudes
#pragma unroll
            for (int rr = 0; rr < 3; rr++) {
                const unsigned* pm = s_mag + (ry - 1 + rr) * MSTR + 2 * s;
                const unsigned a = pm[-1], b0 = pm[0], b1 = pm[1], c = pm[2];
                C[rr][0] = b0; C[rr][1] = b1;
                L[rr][0] = __builtin_amdgcn_alignbit(b0, a, 16);
                R[rr][0] = L[rr][1] = __builtin_amdgcn_alignbit(b1, b0, 16);
                R[rr][1] = __builtin_amdgcn_alignbit(c, b1, 16);
            }
            const unsigned lowp = (unsigned)(iclamp(low, -1, 4095) & 0xffff) * 0x00010001u;
            const unsigned highp = (unsigned)(iclamp(high, -1, 4095) & 0xffff) * 0x00010001u;
            const unsigned high0p = (unsigned)(iclamp(high_main, -1, 4095) & 0xffff) * 0x00010001u;
            unsigned o16[2], o16m[2];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const unsigned cur = C[1][j];
                const unsigned gxp = pg[j], gyp = pg[2 + j];
                const unsigned c_h = pk_gt(cur, L[1][j]) & ~pk_gt(R[1][j], cur);
                const unsigned c_v = pk_gt(cur, C[0][j]) & ~pk_gt(C[2][j], cur);
                // diagonal: signs differ -> (above right, below left), else (above left, below right)
                const unsigned msk = pk_bits(pk_from(gxp ^ gyp) >> 15);
                const unsigned c_d = pk_gt(cur, bsel(msk, R[0][j], L[0][j])) & pk_gt(cur, bsel(msk, L[2][j], R[2][j]));
                const v2u ax = pku_from(pk_bits(pk_abs(pk_from(gxp))));
                const unsigned ay = pk_bits(pk_abs(pk_from(gyp)));
                const v2u q = (ax * (unsigned short)53 + ((ax * (unsigned short)5) >> 8)) >> 7;
                const unsigned s22 = ~pk_gt(ay, pku_bits(q));
                const unsigned s67 = pk_gt(ay, pku_bits(ax + ax + q));
                const unsigned keep = bsel(s22, c_h, bsel(s67, c_v, c_d));
                const unsigned kept = keep & pk_gt(cur, lowp);
                o16[j] = (kept & pk_gt(cur, highp) & 0x00020002u) | (~kept & 0x00010001u);
                o16m[j] = (kept & pk_gt(cur, high0p) & 0x00020002u) | (~kept & 0x00010001u);
            }
            outw = __builtin_amdgcn_perm(o16[1], o16[0], 0x06040200u);
            outw0 = __builtin_amdgcn_perm(o16m[1], o16m[0], 0x06040200u);
        }
        const int off = rowoff(gy, g.pitch) + gx0;
        const unsigned outm = mp0 ? outw0 : outw;                        // the word that is the main Canny's map
        const unsigned edgw = ((outm >> 1) & 0x01010101u) * 0xffu;       // 255 where that byte is 2
        bool wk, wk0 = false;
        if (gx0 + 3 < w) {
            *reinterpret_cast<unsigned*>(mp + off) = outw;
            if (mp0) *reinterpret_cast<unsigned*>(mp0 + off) = outw0;
            if (ep) *reinterpret_cast<unsigned*>(ep + off) = edgw;
            wk = ((outw - 0x01010101u) & ~outw & 0x80808080u) != 0;      // some byte == 0
            if (mp0) wk0 = ((outw0 - 0x01010101u) & ~outw0 & 0x80808080u) != 0;
        } else {
            wk = false;
            for (int q = 0; q < 4 && gx0 + q < w; q++) {
                mp[off + q] = (uint8_t)(outw >> (8 * q)); wk |= ((outw >> (8 * q)) & 0xffu) == 0;
                if (mp0) { mp0[off + q] = (uint8_t)(outw0 >> (8 * q)); wk0 |= ((outw0 >> (8 * q)) & 0xffu) == 0; }
                if (ep) ep[off + q] = (uint8_t)(edgw >> (8 * q));
            }
        }
        if (wk0) s_weak0 = 1;
        if (wk) s_weak = 1;
    }
    __syncthreads();
    if (tid == 0 && s_weak)
        weak_first[1 + atomicAdd(&weak_first[0], 1)] = (int)(((size_t)m_first * g.nb + b) * g.tiles + (size_t)tl.ty * g.tw + tile_x);
    if (tid == 0 && s_weak0)
        weak_main[1 + atomicAdd(&weak_main[0], 1)] = (int)((size_t)b * g.tiles + (size_t)tl.ty * g.tw + tile_x);
    }   // tiles of the group
}

// One hysteresis pass over the tiles of a worklist (all maps of the phase).  A fixed, small grid of workgroups strides over
// the list, one WAVEFRONT per listed 64x32 tile: lane r holds row y0 - 1 + r of the tile (rows -1 and 32 are the read-only
// apron) as two 64-bit masks, S (edge) and W (weak candidate).  One step of "a weak pixel becomes an edge iff an 8-neighbour
// is an edge" is then a handful of 64-bit operations for the whole tile: neighbours in the row are shifts, rows above / below
// come from the neighbouring lanes, and a row is flooded along its weak runs at once by a carry chain
// (((M + S) ^ M) & M, M = W | S, in both bit orders).  The tile reaches its local fixed point in as many steps as its
// longest chain spans ROWS, with no workgroup barrier and no LDS -- scans of photographs, whose contours wander across
// whole tiles, spent a third of their GPU time in the previous byte-per-thread version (up to 300 us per launch).
// The host launches passes back to back; a pass returns at once when the previous pass changed nothing anywhere
// (flags[pass-1] == 0), and from pass 1 on a tile is revisited only if it or one of its 8 neighbours changed in the previous
// pass (chg[tile] == index of the last pass that changed it, + 1).  The result is the unique fixed point, independent of
// scheduling and of the list order.
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

__global__ __launch_bounds__(256) void k_hysteresis(const ImgDesc* __restrict__ desc, Geo g, uint8_t* __restrict__ maps,
                                                    uint8_t* __restrict__ edges, int* __restrict__ flags, int pass,
                                                    const int* __restrict__ wl, int* __restrict__ chg)
{
    static_assert(CT_W == 64 && CT_H + 2 <= 64, "one 64-bit mask per row, one lane per row incl. the apron");
    if (pass > 0 && flags[pass - 1] == 0) return;
    const int nwl = wl[0];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int e = blockIdx.x * 4 + wave; e < nwl; e += gridDim.x * 4) {
        const int key = wl[1 + e];
        const int mb = key / g.tiles, tile = key - mb * g.tiles;       // mb = m * nb + b
        const int b = mb % g.nb;
        const int ty_ = tile / g.tw, tx_ = tile - ty_ * g.tw;
        const int w = desc[b].w, h = desc[b].h;
        const int x0 = tx_ * CT_W, y0 = ty_ * CT_H;
        const size_t tbase = (size_t)mb * g.tiles;
        if (pass > 0) {
            // revisit only if this tile or one of its 8 neighbours changed in the previous pass (lane k looks at neighbour k)
            bool hit = false;
            if (lane < 9) {
                const int ntx = (w + CT_W - 1) / CT_W, nty = (h + CT_H - 1) / CT_H;
                const int tx = tx_ + lane % 3 - 1, ty = ty_ + lane / 3 - 1;
                hit = tx >= 0 && tx < ntx && ty >= 0 && ty < nty && chg[tbase + ty * g.tw + tx] == pass;
            }
            if (__ballot(hit) == 0ull) continue;
        }
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
        if (__ballot(promoted != 0ull) == 0ull) continue;
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
        if (lane == 0) { flags[pass] = 1; chg[tbase + tile] = pass + 1; }
    }
}

}  // namespace i2s
