// Circle erase (img2sgf.py:188-198) fused with the HoughLines accumulation (img2sgf.py:236-244, OpenCV
// hough.cpp HoughLinesStandard), and the accumulator peak search (findLocalMaximums + sort).
#pragma once
#include "i2s_types.h"

namespace i2s {

constexpr int ET_W = 64;
constexpr int ET_H = 64;          // (32 in rounds 1-2: the per-tile set-up and flush cost as much as the pixels)
constexpr int ET_K = ET_H / 16;   // rows per thread: y0 + (tid >> 4) + 16 k
constexpr int LB = 128;          // LDS histogram bins per (call, angle) per tile
constexpr int LANG = 4;          // angle rows reserved per HoughLines call
constexpr int LROWS = 3 * LANG;  // accumulator rows per image

constexpr int TL_CAP = 64;       // circles listed per 64x32 tile; a tile touched by more falls back to scanning all circles

// Concatenate the per-variant circle lists into the reference's `circles` array order (slots of the blur bank,
// img2sgf.py:171-186) and bin the circles' erase boxes by 64x32 tile: tl_cnt[b * g.tiles + tile] (may exceed TL_CAP =
// overflow marker), tl_box[(b * g.tiles + tile) * TL_CAP + k] = everything k_erase_lines needs of the circle -- its box (clipped to
// the image), the centre of its plus and its index -- so that the tile kernel has ONE load between itself and its circles instead of
// the chain count -> index -> circle.  grid (nb), block 256.
struct TlBox { short lo_x, hi_x, lo_y, hi_y, mx, my; unsigned short idx, pad; };
static_assert(sizeof(TlBox) == 16, "one 16-byte load per record");
__global__ __launch_bounds__(256) void k_concat_circles(const ImgDesc* __restrict__ desc, Geo g, const float* __restrict__ vcirc,
                                                        const int* __restrict__ vcount, const int* __restrict__ overflow,
                                                        i2s_result* __restrict__ res, int* __restrict__ tl_cnt, TlBox* __restrict__ tl_box)
{
    __shared__ int s_off[NSLOT + 1];
    const int b = blockIdx.x;
    i2s_result* R = res + b;
    const int w = desc[b].w, h = desc[b].h;
    const int ntx = (w + ET_W - 1) / ET_W, nty = (h + ET_H - 1) / ET_H;
    for (int t = threadIdx.x; t < ntx * nty; t += 256) tl_cnt[(size_t)b * g.tiles + (t / ntx) * g.tw + (t % ntx)] = 0;
    if (threadIdx.x == 0) {
        int o = 0;
        for (int s = 0; s < NSLOT; s++) {
            s_off[s] = o;
            const int n = vcount[b * NVAR + slot_variant(s)];
            R->n_per_slot[s] = n;
            o += n;
        }
        s_off[NSLOT] = o;
        const bool over = overflow[b] != 0 || o > I2S_MAX_CIRCLES;
        R->status = over ? I2S_ST_CAPACITY : 0;
        R->n_circles = over ? 0 : o;
        if (over) s_off[NSLOT] = -1;
    }
    __syncthreads();
    if (s_off[NSLOT] < 0) return;
    // one flat pass over the circles of all slots (a loop over the slots costs ten load -> store -> atomic round trips in a row)
    for (int q = threadIdx.x; q < s_off[NSLOT]; q += 256) {
        int s = 0;
        while (q >= s_off[s + 1]) s++;
        const int i = q - s_off[s];
        const float* src = vcirc + (size_t)(b * NVAR + slot_variant(s)) * g.vcirc_cap * 3;
        const float xc = src[3 * i], yc = src[3 * i + 1], r0 = src[3 * i + 2];
        R->circles[q][0] = xc; R->circles[q][1] = yc; R->circles[q][2] = r0;
        // erase box of circle q: r + 2 in float32, corners rounded half-to-even (img2sgf.py:193-195)
        const float r = r0 + 2.0f;
        const int bx0 = __float2int_rn(xc - r), by0 = __float2int_rn(yc - r);
        const int bx1 = __float2int_rn(xc + r), by1 = __float2int_rn(yc + r);
        const int lo_x = imax(imin(bx0, bx1), 0), hi_x = imin(imax(bx0, bx1), w - 1);
        const int lo_y = imax(imin(by0, by1), 0), hi_y = imin(imax(by0, by1), h - 1);
        if (lo_x > hi_x || lo_y > hi_y) continue;
        TlBox rec;
        rec.lo_x = (short)lo_x; rec.hi_x = (short)hi_x; rec.lo_y = (short)lo_y; rec.hi_y = (short)hi_y;
        rec.mx = (short)iclamp(__float2int_rn(xc), -32768, 32767); rec.my = (short)iclamp(__float2int_rn(yc), -32768, 32767);
        rec.idx = (unsigned short)q; rec.pad = 0;
        for (int ty = lo_y / ET_H; ty <= hi_y / ET_H; ty++)
            for (int tx = lo_x / ET_W; tx <= hi_x / ET_W; tx++) {
                const size_t t = (size_t)b * g.tiles + (size_t)ty * g.tw + tx;
                const int k = atomicAdd(&tl_cnt[t], 1);
                if (k < TL_CAP) tl_box[t * TL_CAP + k] = rec;
            }
    }
}

// grid (tiles_x, tiles_y, nb), block 256, tile 64x32.
// removed(p) = edges(p) if no circle's box covers p; else with i = the LAST circle whose box covers p:
// 255 if p is on circle i's centre plus (cv.circle radius 1), else 0.  (Sequential draw order of
// img2sgf.py:191-198 resolved per pixel: a later rectangle erases earlier dots; each plus lies in its own box.)
// Every non-zero pixel then votes into the rho accumulators of the three HoughLines calls through an LDS
// histogram that is flushed with one global atomic per touched bin.
// lacc[(b * LROWS + c * LANG + n) * lrow + r], r = cvRound(x*cos + y*sin) + (numrho-1)/2, numrho = 2(w+h)+1.
__global__ __launch_bounds__(256) void k_erase_lines(const ImgDesc* __restrict__ desc, Geo g,
                                                     const uint8_t* __restrict__ edges, uint8_t* __restrict__ removed,
                                                     const i2s_result* __restrict__ res, HoughTrig trig,
                                                     int* __restrict__ lacc, int lrow, int gx, int gy,
                                                     const int* __restrict__ tl_cnt, const TlBox* __restrict__ tl_box)
{
    __shared__ TlBox s_box[256];
    __shared__ int s_n;
    __shared__ int s_hist[LROWS][LB];
    __shared__ int s_rmin[LROWS];
    __shared__ unsigned short s_nz[ET_W * ET_H];
    __shared__ float s_cos[LROWS], s_sin[LROWS];
    static_assert(I2S_MAX_CIRCLES <= (1 << 16) && TL_CAP <= 256, "circle index | list position packed into one int");
    const TileId tl = tile_of_block(gx, gy);
    const int b = tl.z;
    const int w = desc[b].w, h = desc[b].h;
    const int x0 = tl.tx * ET_W, y0 = tl.ty * ET_H;
    if (x0 >= w || y0 >= h) return;
    const int tid = threadIdx.x;
    const i2s_result* R = res + b;
    const int nc = R->n_circles;
    const int half = w + h;   // (numrho - 1) / 2
    // this thread's 4 ET_K pixels: 4 consecutive columns x0 + 4 * (tid & 15) .. + 3 on the rows y0 + (tid >> 4) + 16 k (dword
    // loads / stores of the planes; per circle one strip test, then 4 column and 2 row tests)
    const int lx0 = 4 * (tid & 15), ly0 = tid >> 4;
    const uint8_t* e = edges + (size_t)b * g.slot;
    uint8_t* o = removed + (size_t)b * g.slot;
    // every global load of the tile is requested up front (the kernel is a chain of latencies otherwise): the count of the tile's
    // circle list, the list's records WITHOUT waiting for the count (slots past it hold stale bytes, masked below), the edge pixels
    const size_t tslot = (size_t)b * g.tiles + (size_t)tl.ty * g.tw + tl.tx;
    const int tcnt = nc > 0 ? tl_cnt[tslot] : 0;
    TlBox myrec = TlBox{0, 0, 0, 0, 0, 0, 0, 0};
    if (tid < TL_CAP) myrec = tl_box[tslot * TL_CAP + tid];
    unsigned ev2[ET_K];
#pragma unroll
    for (int k = 0; k < ET_K; k++) ev2[k] = 0u;
#pragma unroll
    for (int k = 0; k < ET_K; k++) {
        const int px = x0 + lx0, py = y0 + ly0 + 16 * k;
        if (px < w && py < h) {
            const int off = rowoff(py, g.pitch) + px;
            if (px + 3 < w) ev2[k] = *reinterpret_cast<const unsigned*>(e + off);
            else for (int q = 0; q < 4 && px + q < w; q++) ev2[k] |= (unsigned)e[off + q] << (8 * q);
        }
    }
    for (int i = tid; i < LROWS * LB; i += 256) (&s_hist[0][0])[i] = 0;
    if (tid < LROWS) {
        const int c = tid / LANG, n = tid % LANG;
        int rmin = 0;
        if (n < trig.n[c]) {
            const int xs[2] = {x0, imin(x0 + ET_W, w) - 1}, ys[2] = {y0, imin(y0 + ET_H, h) - 1};
            rmin = 0x7fffffff;
            for (int i = 0; i < 4; i++) {
                const float a = (float)xs[i & 1] * trig.cos_[c][n], bb = (float)ys[i >> 1] * trig.sin_[c][n];
                rmin = imin(rmin, __float2int_rn(a + bb));
            }
        }
        s_rmin[tid] = rmin;
        s_cos[tid] = n < trig.n[c] ? trig.cos_[c][n] : 0.f;
        s_sin[tid] = n < trig.n[c] ? trig.sin_[c][n] : 0.f;
    }
    // best[k][q] = (largest index of a circle whose box covers the pixel) << 8 | its position in s_box, -1 if none
    int best[ET_K][4];
#pragma unroll
    for (int k = 0; k < ET_K; k++)
#pragma unroll
        for (int q = 0; q < 4; q++) best[k][q] = -1;
    auto cover = [&](int n) {
        const int px = x0 + lx0, py = y0 + ly0;
        for (int j = 0; j < n; j++) {
            const int bx0 = s_box[j].lo_x, bx1 = s_box[j].hi_x;
            if (px + 3 < bx0 || px > bx1) continue;
            const int by0 = s_box[j].lo_y, by1 = s_box[j].hi_y, key = ((int)s_box[j].idx << 8) | j;
#pragma unroll
            for (int k = 0; k < ET_K; k++) {
                const int yy = py + 16 * k;
                if (yy < by0 || yy > by1) continue;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (px + q >= bx0 && px + q <= bx1) best[k][q] = imax(best[k][q], key);
            }
        }
    };
    // circles whose erase box touches this tile: the per-tile list built by k_concat_circles (order irrelevant: the
    // largest index wins), or every circle of the image when that list overflowed
    const bool listed = tcnt <= TL_CAP;
    int mxy[ET_K][4];                                      // centre of the deciding circle's plus, packed (overflow path: resolved per chunk)
#pragma unroll
    for (int k = 0; k < ET_K; k++)
#pragma unroll
        for (int q = 0; q < 4; q++) mxy[k][q] = 0;
    if (listed) {
        if (tid < tcnt) s_box[tid] = myrec;
        __syncthreads();
        cover(tcnt);
#pragma unroll
        for (int k = 0; k < ET_K; k++)
#pragma unroll
            for (int q = 0; q < 4; q++)
                if (best[k][q] >= 0) { const TlBox& r = s_box[best[k][q] & 0xff]; mxy[k][q] = ((int)r.mx << 16) | ((int)r.my & 0xffff); }
    } else {
        for (int base = 0; base < nc; base += 256) {
            __syncthreads();
            if (tid == 0) s_n = 0;
            __syncthreads();
            if (base + tid < nc) {
                const int i = base + tid;
                const float xc = R->circles[i][0], yc = R->circles[i][1];
                const float r = R->circles[i][2] + 2.0f;
                const int bx0 = __float2int_rn(xc - r), by0 = __float2int_rn(yc - r);
                const int bx1 = __float2int_rn(xc + r), by1 = __float2int_rn(yc + r);
                const int lo_x = imin(bx0, bx1), hi_x = imax(bx0, bx1), lo_y = imin(by0, by1), hi_y = imax(by0, by1);
                if (hi_x >= x0 && lo_x < x0 + ET_W && hi_y >= y0 && lo_y < y0 + ET_H) {
                    const int k = atomicAdd(&s_n, 1);
                    TlBox rec;
                    rec.lo_x = (short)iclamp(lo_x, -32768, 32767); rec.hi_x = (short)iclamp(hi_x, -32768, 32767);
                    rec.lo_y = (short)iclamp(lo_y, -32768, 32767); rec.hi_y = (short)iclamp(hi_y, -32768, 32767);
                    rec.mx = (short)iclamp(__float2int_rn(xc), -32768, 32767); rec.my = (short)iclamp(__float2int_rn(yc), -32768, 32767);
                    rec.idx = (unsigned short)i; rec.pad = 0;
                    s_box[k] = rec;
                }
            }
            __syncthreads();
            int before[ET_K][4];
#pragma unroll
            for (int k = 0; k < ET_K; k++)
#pragma unroll
                for (int q = 0; q < 4; q++) before[k][q] = best[k][q];
            cover(s_n);
#pragma unroll
            for (int k = 0; k < ET_K; k++)
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (best[k][q] != before[k][q]) { const TlBox& r = s_box[best[k][q] & 0xff]; mxy[k][q] = ((int)r.mx << 16) | ((int)r.my & 0xffff); }
        }
    }
    __syncthreads();                                    // s_hist / s_rmin initialisation; everybody is done with s_box
    if (tid == 0) s_n = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ET_K; k++) {
        const int ly = ly0 + 16 * k;
        const int px = x0 + lx0, py = y0 + ly;
        if (px >= w || py >= h) continue;
        const int off = rowoff(py, g.pitch) + px;
        const bool whole = px + 3 < w;
        const unsigned ev = ev2[k];
        unsigned outw = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            unsigned val = (ev >> (8 * q)) & 0xffu;
            if (best[k][q] >= 0) {
                const int mx = mxy[k][q] >> 16, my = (int)(short)(mxy[k][q] & 0xffff);
                const bool plus = (px + q == mx && iabs_(py - my) <= 1) || (py == my && iabs_(px + q - mx) <= 1);
                val = plus ? 255u : 0u;
            }
            outw |= val << (8 * q);
            if (val != 0 && px + q < w) s_nz[atomicAdd(&s_n, 1)] = (unsigned short)(ly * ET_W + lx0 + q);   // non-zero pixels are few: vote on a dense list
        }
        if (whole) *reinterpret_cast<unsigned*>(o + off) = outw;
        else for (int q = 0; q < 4 && px + q < w; q++) o[off + q] = (uint8_t)(outw >> (8 * q));
    }
    __syncthreads();
    {
        // every non-zero pixel votes once per angle of the three HoughLines calls: threads stride over the pixels, the angle
        // rows are a uniform inner loop (no per-item division by the angle count, no select chain over the row table)
        const int nnz = s_n;
        for (int pi = tid; pi < nnz; pi += 256) {
            const int p = s_nz[pi];
            const float fx = (float)(x0 + (p & (ET_W - 1))), fy = (float)(y0 + p / ET_W);
#pragma unroll
            for (int c = 0; c < 3; c++)
                for (int n = 0; n < trig.n[c]; n++) {
                    const int row = c * LANG + n;
                    const float a = fx * s_cos[row], bb = fy * s_sin[row];
                    const int r = __float2int_rn(a + bb);
                    const unsigned bin = (unsigned)(r - s_rmin[row]);
                    if (bin < (unsigned)LB) atomicAdd(&s_hist[row][bin], 1);
                    else atomicAdd(&lacc[((size_t)b * LROWS + row) * lrow + r + half], 1);
                }
        }
    }
    __syncthreads();
    for (int i = tid; i < LROWS * LB; i += 256) {
        const int row = i / LB, bin = i - row * LB;
        const int cnt = s_hist[row][bin];
        if (cnt) atomicAdd(&lacc[((size_t)b * LROWS + row) * lrow + s_rmin[row] + bin + half], cnt);
    }
}

// Peaks of one HoughLines call (findLocalMaximums + std::sort(hough_cmp_gt)), appended to out[] as rho.
// Runs inside an LP_THREADS block; s_key is LDS scratch of I2S_MAX_LINES entries.  Returns the count (or -1
// on overflow) in *s_cnt after the final __syncthreads().
// The accumulator rows are scanned along rho (consecutive lanes, consecutive counters), four counters per thread in flight; the
// neighbours are only fetched for the few counters above the threshold.  The order in which peaks are found does not matter: they
// are ranked by (count, OpenCV's padded index) afterwards.
constexpr int LP_THREADS = 1024;
__device__ __forceinline__ void lines_peaks_call(const int* __restrict__ acc /* call base: LANG rows */, int lrow, int numrho,
                                                 int numangle, int threshold, bool negate, unsigned long long* s_key,
                                                 int* s_cnt, float* __restrict__ out, int out_base)
{
    const int tid = threadIdx.x;
    if (tid == 0) *s_cnt = 0;
    __syncthreads();
    for (int n = 0; n < numangle; n++) {
        const int* row = acc + (size_t)n * lrow;
        for (int r0 = tid; r0 < numrho; r0 += 4 * LP_THREADS) {
            int a4[4];
#pragma unroll
            for (int q = 0; q < 4; q++) { const int r = r0 + q * LP_THREADS; a4[q] = row[imin(r, numrho - 1)]; }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int r = r0 + q * LP_THREADS, a = a4[q];
                if (r >= numrho || a <= threshold) continue;
                const int left = r > 0 ? row[r - 1] : 0;
                const int right = r < numrho - 1 ? row[r + 1] : 0;
                const int up = n > 0 ? row[r - lrow] : 0;
                const int down = n < numangle - 1 ? row[r + lrow] : 0;
                if (a > left && a >= right && a > up && a >= down) {
                    const int k = atomicAdd(s_cnt, 1);
                    const unsigned idx = (unsigned)((n + 1) * (numrho + 2) + r + 1);   // OpenCV's padded index: the sort tie-break
                    if (k < I2S_MAX_LINES) s_key[k] = ((unsigned long long)(0x7fffffffu - (unsigned)a) << 32) | idx;
                }
            }
        }
    }
    __syncthreads();
    const int cnt = *s_cnt;
    if (cnt > I2S_MAX_LINES) { __syncthreads(); if (tid == 0) *s_cnt = -1; __syncthreads(); return; }
    for (int i = tid; i < cnt; i += LP_THREADS) {
        const unsigned long long k = s_key[i];
        int rank = 0;
        for (int j = 0; j < cnt; j++) rank += (s_key[j] < k) ? 1 : 0;   // keys are unique (idx)
        const int idx = (int)(k & 0xffffffffu);
        const int n = idx / (numrho + 2) - 1;
        const int r = idx - (n + 1) * (numrho + 2) - 1;
        const float rho = ((float)r - (float)(numrho - 1) * 0.5f) * 1.0f;
        if (out_base + rank < I2S_MAX_LINES) out[out_base + rank] = negate ? -rho : rho;
    }
    __syncthreads();
}

// find_all_lines (img2sgf.py:258-265): hlines from call 0; vlines = [call 1 ; call 2 with rho negated] (245-251).
// grid (nb), block LP_THREADS.
__global__ __launch_bounds__(LP_THREADS) void k_line_peaks(const ImgDesc* __restrict__ desc, const int* __restrict__ lacc, int lrow,
                                                    HoughTrig trig, i2s_result* __restrict__ res)
{
    __shared__ unsigned long long s_key[I2S_MAX_LINES];
    __shared__ int s_cnt;
    const int b = blockIdx.x;
    i2s_result* R = res + b;
    const int w = desc[b].w, h = desc[b].h, thr = desc[b].line_thr;
    const int numrho = 2 * (w + h) + 1;
    const int* acc = lacc + (size_t)b * LROWS * lrow;
    bool over = false;
    lines_peaks_call(acc, lrow, numrho, trig.n[0], thr, false, s_key, &s_cnt, R->hlines, 0);
    const int nh = s_cnt;
    __syncthreads();
    lines_peaks_call(acc + (size_t)LANG * lrow, lrow, numrho, trig.n[1], thr, false, s_key, &s_cnt, R->vlines, 0);
    const int nv1 = s_cnt;
    __syncthreads();
    over = nh < 0 || nv1 < 0;
    int nv2 = 0;
    if (!over) {
        lines_peaks_call(acc + (size_t)2 * LANG * lrow, lrow, numrho, trig.n[2], thr, true, s_key, &s_cnt, R->vlines, nv1);
        nv2 = s_cnt;
        __syncthreads();
        over = nv2 < 0 || nv1 + nv2 > I2S_MAX_LINES;
    }
    if (threadIdx.x == 0) {
        R->line_threshold = thr;
        if (over) { R->status = I2S_ST_CAPACITY; R->n_hlines = 0; R->n_vlines = 0; }
        else { R->n_hlines = nh; R->n_vlines = nv1 + nv2; }
    }
}

}  // namespace i2s
