// find_grid() after the Hough transforms (img2sgf.py:546-576): clustering of the rho lists (268-292), grid
// repair / validation (335-445), circle snapping (448-465), the mean-intensity stone classifier (468-543),
// side-to-move guess and board alignment (484-494, 529-534).  One workgroup per image; the O(10^2) scalar
// float64 logic runs on lane 0 with exactly the reference's operation order (Python float == IEEE double,
// round() == rint), the window sums of the classifier run one wavefront per stone.
#pragma once
#include "i2s_types.h"

namespace i2s {

constexpr int GRID_THREADS = 1024;   // 16 wavefronts: the classifier sums 16 stone windows at a time

struct GridParams {
    double min_grid_spacing, big_space_ratio;
    int black_threshold, align_x, align_y, pad;
};

// img2sgf.py:400-417.  x[0..n) -> possibly shifted window; returns new length, *first = start offset.
__device__ __host__ inline int truncate_grid(int n, int* first)
{
    if (n == I2S_BOARD_SIZE + 2) { *first += 1; return n - 2; }
    if (n == I2S_BOARD_SIZE + 1) return n - 1;
    return n;
}

// img2sgf.py:335-397.  in: x[0..n) sorted.  out: y[0..*m).  Returns 0 on success or an I2S_ST_H_* code
// (caller adds the axis offset).  y may hold up to max(n, BOARD_SIZE + 3) values.
__device__ __host__ inline int complete_grid(const double* x, int n, double* y, int* m, double min_spacing, double ratio)
{
    if (n == 0) return I2S_ST_H_NO_LINES;
    if (n == 1) return I2S_ST_H_ONE_LINE;
    double min_space = x[1] - x[0];
    for (int i = 1; i < n - 1; i++) { const double s = x[i + 1] - x[i]; if (s < min_space) min_space = s; }
    if (min_space < min_spacing) return I2S_ST_H_TOO_CLOSE;
    const double bound = min_space * ratio;
    int nbig = 0, nsmall = 0;
    double max_space = 0;
    bool have_small = false;
    for (int i = 0; i < n - 1; i++) {
        const double s = x[i + 1] - x[i];
        if (s > bound) nbig++;
        else { nsmall++; if (!have_small || s > max_space) { max_space = s; have_small = true; } }
    }
    if (nbig == 0) { for (int i = 0; i < n; i++) y[i] = x[i]; *m = n; return 0; }
    const double average_space = (min_space + max_space) / 2;
    int cnt = nsmall;
    for (int i = 0; i < n - 1; i++) {
        const double s = x[i + 1] - x[i];
        if (s > bound) cnt += (int)rint(s / average_space);
    }
    if (cnt > I2S_BOARD_SIZE + 2) return I2S_ST_H_TOO_WIDE;
    cnt += 1;
    if (n < cnt) {
        y[0] = x[0];
        int i = 1, j = 1;
        for (int q = 0; q < n - 1; q++) {
            const double s = x[q + 1] - x[q];
            if (s <= max_space) { y[i] = x[j]; i++; j++; }
            else {
                const int mm = (int)rint(s / average_space);
                for (int k = 0; k < mm; k++) { y[i] = x[j - 1] + (double)(k + 1) * s / (double)mm; i++; }
                j++;
            }
        }
        *m = cnt;
        return 0;
    }
    for (int i = 0; i < n; i++) y[i] = x[i];
    *m = n;
    return 0;
}

// img2sgf.py:448-459 with bisect_left.
__device__ __host__ inline int closest_index(double a, const double* x, int n)
{
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) / 2; if (x[mid] < a) lo = mid + 1; else hi = mid; }
    const int i = lo;
    if (i == 0) return 0;
    if (i == n) return i - 1;
    return (a - x[i - 1] <= x[i] - a) ? i - 1 : i;
}

// Sort-and-cluster of one rho list (find_clusters_fixed_threshold + get_cluster_centres, 268-292):
// single linkage with distance_threshold d on a 1-D set == split the sorted values where the gap is >= d;
// centre = float32 mean (sums of integer-valued rho are exact).  s_sorted: LDS scratch.  Whole block calls it.
// Result in out[0..*n_out) (ascending), *n_out = -1 on capacity overflow.
__device__ __forceinline__ void cluster_axis(const float* __restrict__ rho, int n, float* s_sorted, double min_spacing,
                                             double* out, int* n_out)
{
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += GRID_THREADS) {
        const float v = rho[i];
        int rank = 0;
        for (int j = 0; j < n; j++) { const float u = rho[j]; rank += (u < v || (u == v && j < i)) ? 1 : 0; }
        s_sorted[rank] = v;
    }
    __syncthreads();
    if (tid == 0) {
        int k = 0;
        if (n >= 2) {
            int start = 0;
            for (int i = 1; i <= n; i++) {
                if (i == n || (double)s_sorted[i] - (double)s_sorted[i - 1] >= min_spacing) {
                    float sum = 0.f;
                    for (int q = start; q < i; q++) sum += s_sorted[q];
                    const float mean = sum / (float)(i - start);
                    if (k < I2S_MAX_CENTRES) out[k] = (double)mean;
                    k++;
                    start = i;
                }
            }
        }
        *n_out = k > I2S_MAX_CENTRES ? -1 : k;
    }
    __syncthreads();
}

// grid (nb), block GRID_THREADS.  grey = variant plane 0.  Reads res[b].{circles, n_circles, hlines, vlines, status}
// and fills the rest of res[b] and boards[b].  do_cluster = 0 re-runs only identify_board on the stored grid
// (apply_black_thresh, img2sgf.py:762-766).  do_cluster = 2 is validate_grid alone (:420-445, i2s_validate_grid): the cluster
// centres are taken from res[b].{hcentres, vcentres} as they are -- float64, any spacing, in the given order -- and the kernel
// stops after the radius filter (no grey plane is looked at).
__global__ __launch_bounds__(1024) void k_grid(const ImgDesc* __restrict__ desc, Geo g,
                                              GridParams gp, int do_cluster, i2s_result* __restrict__ res,
                                              i2s_board* __restrict__ boards)
{
    static_assert(I2S_MAX_CENTRES >= I2S_MAX_LINES, "every cluster holds a line: the centres cannot overflow before the lines do");
    __shared__ float s_sorted[I2S_MAX_LINES];
    __shared__ double s_tmp[2][I2S_MAX_CENTRES + 4];
    __shared__ double s_cen[2][I2S_MAX_CENTRES];      // hcentres / vcentres
    __shared__ double s_cmp[2][I2S_MAX_CENTRES];      // hcentres_complete / vcentres_complete (global memory is only written)
    __shared__ int s_win[I2S_BOARD_SIZE * I2S_BOARD_SIZE][4];
    __shared__ unsigned char s_det[I2S_BOARD_SIZE][I2S_BOARD_SIZE];
    __shared__ double s_br[I2S_BOARD_SIZE * I2S_BOARD_SIZE];
    __shared__ int s_i[8];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int w = desc[b].w, h = desc[b].h;
    i2s_result* R = res + b;
    i2s_board* B = boards + b;
    const bool capacity = R->status == I2S_ST_CAPACITY;
    if (capacity) {
        // nothing of an overflowed image is valid: the record says so in every count, and neither board keeps what an earlier pass
        // left in this slot (the circle count too -- k_line_peaks' overflow leaves it standing, and the packed full record is laid
        // out from the board record's count)
        for (int i = tid; i < I2S_BOARD_SIZE * I2S_BOARD_SIZE; i += GRID_THREADS) {
            (&B->board[0][0])[i] = 0; (&R->board[0][0])[i] = 0; (&R->detected[0][0])[i] = 0;
        }
        if (tid == 0) {
            R->found_grid = R->valid_grid = R->board_ready = 0; R->hsize = R->vsize = 0;
            R->n_hcentres = R->n_vcentres = R->n_hcomplete = R->n_vcomplete = 0;
            R->n_stones = R->n_black = R->n_white = R->side_to_move = 0; R->n_circles_kept = 0; R->n_circles = 0;
            B->status = I2S_ST_CAPACITY; B->side_to_move = 0; B->hsize = B->vsize = 0; B->found_grid = B->valid_grid = 0;
            B->n_black = B->n_white = 0; B->n_circles = 0; B->line_threshold = (uint16_t)R->line_threshold;
        }
        return;
    }
    if (do_cluster) {
        if (do_cluster == 2) {
            for (int i = tid; i < R->n_hcentres; i += GRID_THREADS) s_cen[0][i] = R->hcentres[i];
            for (int i = tid; i < R->n_vcentres; i += GRID_THREADS) s_cen[1][i] = R->vcentres[i];
            if (tid == 0) { s_i[0] = R->n_hcentres; s_i[1] = R->n_vcentres; }
            __syncthreads();
        } else {
            cluster_axis(R->hlines, R->n_hlines, s_sorted, gp.min_grid_spacing, s_cen[0], &s_i[0]);
            cluster_axis(R->vlines, R->n_vlines, s_sorted, gp.min_grid_spacing, s_cen[1], &s_i[1]);
        }
        if (tid == 0) {
            int status = 0;
            const int nh = s_i[0], nv = s_i[1];
            if (nh < 0 || nv < 0) status = I2S_ST_CAPACITY;
            R->n_hcentres = nh < 0 ? 0 : nh; R->n_vcentres = nv < 0 ? 0 : nv;
            R->found_grid = (nh > 0 && nv > 0) ? 1 : 0;
            R->valid_grid = 0; R->board_ready = 0; R->hsize = 0; R->vsize = 0;
            R->n_hcomplete = 0; R->n_vcomplete = 0; R->hspace = 0; R->vspace = 0;
            s_i[4] = 0;                                            // 1 -> apply the radius filter (valid grid)
            if (status == 0) {
                // validate_grid (:420-445): horizontal lines first
                int first = 0, m = 0;
                int n = truncate_grid(nh, &first);
                int rc = complete_grid(s_cen[0] + first, n, s_tmp[0], &m, gp.min_grid_spacing, gp.big_space_ratio);
                if (rc) status = rc;
                else {
                    int f2 = 0;
                    const int mh = truncate_grid(m, &f2);
                    for (int i = 0; i < mh; i++) s_cmp[0][i] = s_tmp[0][f2 + i];
                    first = 0;
                    n = truncate_grid(nv, &first);
                    rc = complete_grid(s_cen[1] + first, n, s_tmp[1], &m, gp.min_grid_spacing, gp.big_space_ratio);
                    if (rc) status = rc + (I2S_ST_V_NO_LINES - I2S_ST_H_NO_LINES);
                    else {
                        f2 = 0;
                        const int mv = truncate_grid(m, &f2);
                        for (int i = 0; i < mv; i++) s_cmp[1][i] = s_tmp[1][f2 + i];
                        const int vsize = mh, hsize = mv;    // number of horizontal lines = vertical size (:435-436)
                        const double hspace = (s_cmp[0][mh - 1] - s_cmp[0][0]) / (double)vsize;
                        const double vspace = (s_cmp[1][mv - 1] - s_cmp[1][0]) / (double)hsize;
                        s_tmp[0][0] = (hspace < vspace ? hspace : vspace) * 0.3;     // min_circle_size (:441)
                        s_tmp[0][1] = (hspace > vspace ? hspace : vspace) * 0.65;    // max_circle_size (:442)
                        s_i[4] = 1;
                        R->valid_grid = 1; R->vsize = vsize; R->hsize = hsize;
                        R->n_hcomplete = mh; R->n_vcomplete = mv; R->hspace = hspace; R->vspace = vspace;
                        if (hsize > I2S_BOARD_SIZE) status = I2S_ST_TOO_MANY_VLINES;
                        else if (vsize > I2S_BOARD_SIZE) status = I2S_ST_TOO_MANY_HLINES;
                    }
                }
            }
            R->status = status;
            s_i[5] = 0;
        }
        __syncthreads();
        for (int i = tid; i < R->n_hcentres; i += GRID_THREADS) R->hcentres[i] = s_cen[0][i];
        for (int i = tid; i < R->n_vcentres; i += GRID_THREADS) R->vcentres[i] = s_cen[1][i];
        for (int i = tid; i < R->n_hcomplete; i += GRID_THREADS) R->hcentres_complete[i] = s_cmp[0][i];
        for (int i = tid; i < R->n_vcomplete; i += GRID_THREADS) R->vcentres_complete[i] = s_cmp[1][i];
        {
            // radius filter (:443), all threads; a failed validation returns `circles` unfiltered (:426)
            const bool filt = s_i[4] != 0;
            const double lo = s_tmp[0][0], hi = s_tmp[0][1];
            int mine = 0;
            for (int i = tid; i < R->n_circles; i += GRID_THREADS) {
                int keep = 1;
                if (filt) { const double r = (double)R->circles[i][2]; keep = (lo < r && r < hi) ? 1 : 0; }
                R->circle_kept[i] = (uint8_t)keep;
                mine += keep;
            }
            if (mine) atomicAdd(&s_i[5], mine);
        }
        __syncthreads();
        if (tid == 0) R->n_circles_kept = s_i[5];
        __syncthreads();
        if (do_cluster == 2) return;
    }
    if (!do_cluster) {
        for (int i = tid; i < R->n_hcomplete; i += GRID_THREADS) s_cmp[0][i] = R->hcentres_complete[i];
        for (int i = tid; i < R->n_vcomplete; i += GRID_THREADS) s_cmp[1][i] = R->vcentres_complete[i];
        __syncthreads();
    }
    // identify_board (:497-543)
    const bool ready = R->valid_grid && R->hsize <= I2S_BOARD_SIZE && R->vsize <= I2S_BOARD_SIZE && R->status != I2S_ST_CAPACITY;
    const int hsize = R->hsize, vsize = R->vsize;
    for (int i = tid; i < I2S_BOARD_SIZE * I2S_BOARD_SIZE; i += GRID_THREADS) (&s_det[0][0])[i] = 0;
    __syncthreads();
    if (ready) {
        for (int i = tid; i < R->n_circles; i += GRID_THREADS) {
            if (!R->circle_kept[i]) continue;
            const int ci = closest_index((double)R->circles[i][0], s_cmp[1], hsize);
            const int cj = closest_index((double)R->circles[i][1], s_cmp[0], vsize);
            s_det[ci][cj] = I2S_STONE;
        }
    }
    __syncthreads();
    // the stones in the reference's order (j over hsize, k over vsize, :510-514): thread t looks at cell (t / vsize, t % vsize); a
    // stone's position in the list is the number of stones in front of it (wave ballots + the waves' totals) -- the loop over the
    // 361 cells on one lane, with its float64 window arithmetic, used to be most of this kernel's time
    __shared__ int s_wtot[GRID_THREADS / 64];
    int my_rank = -1, my_j = 0, my_k = 0;
    {
        const int ncell = ready ? hsize * vsize : 0;
        bool stone = false;
        if (tid < ncell) { my_j = tid / vsize; my_k = tid - my_j * vsize; stone = s_det[my_j][my_k] == I2S_STONE; }
        const unsigned long long m = __ballot(stone);
        const int wv = tid >> 6, ln = tid & 63;
        if (ln == 0) s_wtot[wv] = __popcll(m);
        __syncthreads();
        int before = 0, total = 0;
        for (int q = 0; q < GRID_THREADS / 64; q++) { const int c = s_wtot[q]; total += c; if (q < wv) before += c; }
        if (stone) {
            my_rank = before + __popcll(m & ((1ull << ln) - 1ull));
            // average_intensity (:468-481); note x uses hspace, y uses vspace (reference quirk)
            const double hspace = R->hspace, vspace = R->vspace;
            const double x = s_cmp[1][my_j], y = s_cmp[0][my_k];
            int xmin = (int)rint(x - hspace / 2), xmax = (int)rint(x + hspace / 2);
            int ymin = (int)rint(y - vspace / 2), ymax = (int)rint(y + vspace / 2);
            xmin = imax(0, xmin); ymin = imax(0, ymin); xmax = imin(w, xmax); ymax = imin(h, ymax);
            s_win[my_rank][0] = xmin; s_win[my_rank][1] = xmax; s_win[my_rank][2] = ymin; s_win[my_rank][3] = ymax;
        }
        if (tid == 0) s_i[2] = total;
    }
    __syncthreads();
    const int ns = s_i[2];
    {
        const int wave = tid >> 6, lane = tid & 63;
        const uint8_t* gp0 = desc[b].grey;                               // may be the source image itself (ImgDesc::grey)
        const int gpitch = desc[b].gpitch;
        const uint8_t* gend = gp0 + rowoff(desc[b].h - 1, gpitch) + desc[b].w;      // one past the last pixel
        for (int s0 = 0; s0 < ns; s0 += GRID_THREADS / 64) {
            const int s = s0 + wave;
            unsigned sum = 0;
            int cnt = 0;
            if (s < ns) {
                const int xmin = s_win[s][0], xmax = s_win[s][1], ymin = s_win[s][2], ymax = s_win[s][3];
                const int bw = imax(xmax - xmin, 0), bh = imax(ymax - ymin, 0);
                cnt = bw * bh;
                // 4 pixels per lane (one unaligned dword, bytes beyond the window masked off, v_sad_u8 adds the four bytes),
                // L lanes per row, 64 / L rows per step; a dword may reach 3 bytes past the window, never past the plane's
                // allocation (rows are padded to the pitch, the plane array ends with slack)
                const int L = imin(64, (bw + 3) >> 2);
                const int rows_per = 64 / imax(L, 1);
                const int ry = lane / imax(L, 1), lx = lane - ry * L;
                // the plane is cold by now (each dword is an HBM round trip), so a lane keeps four rows' loads in flight: the
                // address is clamped into the window and the value dropped afterwards, which keeps the loads free of branches
                auto ld = [&](int yy, int xx) -> unsigned {
                    const int yc = imin(yy, bh - 1);
                    const uint8_t* p = gp0 + rowoff(ymin + yc, gpitch) + xmin + xx;
                    const long over = (long)(p + 4 - gend);                  // > 0: the last pixels of an image used in place
                    unsigned v4;
                    __builtin_memcpy(&v4, over > 0 ? gend - 4 : p, 4);
                    if (over > 0) v4 >>= 8 * (int)over;                      // over <= 3: p is a pixel of the plane
                    const int nvalid = bw - xx;                              // >= 1
                    if (nvalid < 4) v4 &= (1u << (8 * nvalid)) - 1u;
                    return yy < bh ? v4 : 0u;
                };
                if (bw > 0 && ry < rows_per)
                    for (int xx = 4 * lx; xx < bw; xx += 4 * L)
                        for (int yy = ry; yy < bh; yy += 4 * rows_per) {
                            const unsigned u0 = ld(yy, xx), u1 = ld(yy + rows_per, xx);
                            const unsigned u2 = ld(yy + 2 * rows_per, xx), u3 = ld(yy + 3 * rows_per, xx);
                            sum = __builtin_amdgcn_sad_u8(u0, 0u, sum);
                            sum = __builtin_amdgcn_sad_u8(u1, 0u, sum);
                            sum = __builtin_amdgcn_sad_u8(u2, 0u, sum);
                            sum = __builtin_amdgcn_sad_u8(u3, 0u, sum);
                        }
            }
            for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
            if (s < ns && lane == 0) {
                // np.mean of uint8: float64 sum / count; empty slice -> nan (compares False -> WHITE)
                s_br[s] = cnt > 0 ? (double)sum / (double)cnt : __builtin_nan("");
            }
        }
    }
    __syncthreads();
    if (tid == 0) s_i[6] = 0;
    __syncthreads();
    if (my_rank >= 0) {
        const bool black = s_br[my_rank] <= (double)gp.black_threshold;   // NaN (an empty window) compares false: white
        s_det[my_j][my_k] = black ? I2S_BLACK : I2S_WHITE;
        if (black) atomicAdd(&s_i[6], 1);
    }
    __syncthreads();
    if (tid == 0) {
        const int nblack = s_i[6];
        const int nwhite = ns - nblack;
        R->n_stones = ns; R->n_black = nblack; R->n_white = nwhite;
        R->board_ready = ready ? 1 : 0;
        R->side_to_move = ready ? (nblack <= nwhite ? 1 : 2) : 0;
        s_i[3] = ready ? 1 : 0;
    }
    __syncthreads();
    for (int i = tid; i < ns; i += GRID_THREADS) R->brightness[i] = s_br[i];
    // align_board (:484-494) + publish
    const int xoff = (gp.align_x == I2S_ALIGN_RIGHT) ? I2S_BOARD_SIZE - hsize : 0;
    const int yoff = (gp.align_y == I2S_ALIGN_BOTTOM) ? I2S_BOARD_SIZE - vsize : 0;
    for (int i = tid; i < I2S_BOARD_SIZE * I2S_BOARD_SIZE; i += GRID_THREADS) {
        const int bi = i / I2S_BOARD_SIZE, bj = i - bi * I2S_BOARD_SIZE;
        uint8_t v = 0;
        if (s_i[3]) {
            const int di = bi - xoff, dj = bj - yoff;
            if (di >= 0 && di < hsize && dj >= 0 && dj < vsize) v = s_det[di][dj];
        }
        R->board[bi][bj] = v;
        B->board[bi][bj] = v;
        R->detected[bi][bj] = (s_i[3] && bi < hsize && bj < vsize) ? s_det[bi][bj] : 0;
    }
    if (tid == 0) {
        B->status = (uint8_t)R->status; B->side_to_move = (uint8_t)R->side_to_move;
        B->hsize = (uint8_t)imin(R->hsize, 255); B->vsize = (uint8_t)imin(R->vsize, 255);
        B->found_grid = (uint8_t)R->found_grid; B->valid_grid = (uint8_t)R->valid_grid; B->pad0 = 0;
        B->n_black = (uint16_t)R->n_black; B->n_white = (uint16_t)R->n_white;
        B->n_circles = (uint16_t)R->n_circles; B->line_threshold = (uint16_t)R->line_threshold;
        for (int i = 0; i < 8; i++) B->pad[i] = 0;
    }
}

}  // namespace i2s
