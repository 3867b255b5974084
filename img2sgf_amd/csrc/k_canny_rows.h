// Sobel + non-maximum suppression of single-channel planes, register-resident (round 2): the Canny of img2sgf.py:162 for grey
// sources and the Canny inside every cv.HoughCircles call (:180), same arithmetic as round 1's LDS-tiled kernel (OpenCV
// canny.cpp: Sobel 3x3 CV_16S with BORDER_REPLICATE, L1 magnitude, TG22 sectors, strict '>' thresholds), no LDS, no barrier.
// A lane owns one dword column (4 pixels) and walks down CR_R output rows; a wavefront is 256 pixels of a row.  Per input
// row: the neighbour dwords come from the adjacent lanes (DPP), the pixels become 16-bit pairs, the vertical sums / differences
// of the 3-row ring give the gradients of the row above (two pixels per packed instruction), and once the magnitudes of three
// rows exist the row between them is suppressed and written.  The magnitudes left and right of the lane's 4 pixels come from
// the neighbour lanes; at the two ends of the wavefront -- where the neighbour pixel belongs to another wavefront -- lanes 0
// and 63 compute that one extra magnitude themselves from the three columns around it (byte dot products).
// Map values: 0 = weak candidate, 1 = no edge, 2 = edge.  Tiles (64 x 32) that hold weak pixels go to the hysteresis worklist.
#pragma once
#include "k_canny.h"
#include "k_filters.h"

namespace i2s {

constexpr int CR_R = CT_H;       // output rows per wavefront = one row of hysteresis tiles (+4 apron rows of input; 4 x CT_H measured: no faster)

struct CrThr { unsigned lowp1, highp1, high0p1; };      // thresholds + 1, twice (16-bit halves): m > thr  <=>  m >= thr + 1

// suppression of one pixel pair (two 16-bit halves): cur = magnitudes, L / R / above / below and the four diagonals, ax / ay the
// absolute gradients, sgf = bit 15 set where the gradient signs differ.  Returns the map values (0 / 1 / 2 per half) for the
// thresholds (low, high) and -- TWO -- (low, high0).
// Every comparison is (a | 0x8000) - b in ONE 32-bit subtraction (a full-rate instruction, the packed 16-bit forms are not): all values
// are below 2^15, so no borrow crosses the halves and bit 15 of each half says a >= b.  The answers stay in bit 15: the sector picks its
// pair of neighbours by v_bitop3 -- horizontal (left, right), vertical (above, below), diagonal by the gradient signs: signs differ ->
// (above right, below left), else (above left, below right).  OpenCV keeps m > first && m >= second on the axes, m > both on a diagonal.
template <bool TWO>
__device__ __forceinline__ void cr_nms_pair(unsigned cur, unsigned l1, unsigned r1, unsigned c0, unsigned c2, unsigned l0, unsigned r0,
                                            unsigned l2, unsigned r2, unsigned axp, unsigned ay, unsigned sgf, const CrThr& th, unsigned& o, unsigned& om)
{
    constexpr unsigned H = 0x80008000u, ONE = 0x00010001u;
    const v2u ax = pku_from(axp);
    // |dy| 2^15 < |dx| 13573  <=>  |dy| <= q,  |dy| 2^15 > |dx| 79109  <=>  |dy| > 2 |dx| + q,  q = floor(|dx| 13573 / 2^15) = (|dx| 53 + (|dx| 5 >> 8)) >> 7:
    // 13573 = 53 * 256 + 5 is odd, so the quotient is never exact for |dx| > 0, and for |dx| = |dy| = 0 the magnitude is 0 and nothing is kept
    const unsigned q = pku_bits((ax * (unsigned short)53 + ((ax * (unsigned short)5) >> 8)) >> 7);
    const unsigned s22 = (q | H) - ay;
    const unsigned s67 = (ay | H) - (axp + axp + q + ONE);
    const unsigned curh = cur | H, cur1 = curh - ONE;                  // (cur1 - b): cur > b
    const unsigned kh = bitop3<0x80>(s22, cur1 - l1, curh - r1);
    const unsigned kv = bitop3<0x80>(s67, cur1 - c0, curh - c2);
    const unsigned kd = bitop3<0xCA>(sgf, (cur1 - r0) & (cur1 - l2), (cur1 - l0) & (cur1 - r2));
    const unsigned k = bitop3<0xFE>(kh, kv, bitop3<0x02>(s22, s67, kd)) & (curh - th.lowp1);
    const unsigned nk = ((k ^ H) >> 15) & ONE;
    o = nk | (((k & (curh - th.highp1)) >> 14) & 0x00020002u);
    if (TWO) om = nk | (((k & (curh - th.high0p1)) >> 14) & 0x00020002u);
}

// main_mode: 0 = HoughCircles' Canny of variants v_first .. (plane v -> map 1 + v), 1 = main Canny of grey
// sources (plane 0 -> map 0 + edge image, threshold high_main), 2 = both at once for the grey plane, 3 = main Canny of COLOURED
// sources (round 4): the R, G, B planes k_split_rgb wrote (`planes` + c * nb * slot), per pixel the gradient of the channel with the
// largest L1 magnitude, ties to the lower index (OpenCV canny.cpp, Appendix A.2 step 2) -- the Sobel part three times, everything
// behind it once; images whose channels are equal everywhere (has_colour == 0) have been done in mode 1 / 2.
// grid: ceil(w / 1024) x ceil(h / CR_R) x (nb * variants) workgroups of 4 wavefronts (4 consecutive 256-pixel column groups).
//
// BIN (round 4): planes whose pixels are all 0 or 255 in the band a wavefront walks -- the clean diagrams' grey plane and medians, every
// edge image -- take the walk in BYTES first: with pixels as 0 / 1 the Sobel sums fit 4 bits (|dx|, |dy| <= 4 units of 255, magnitude <= 8),
// so a lane's four pixels stay in one register from the load to the map byte (SWAR adds that cannot carry across bytes, compares through
// bit 7, the sector limits as two 5-entry v_perm tables, selects as single v_bitop3).  The two-valued test runs on the rows as they are
// loaded; a band that fails it is walked again by the 16-bit code below, which overwrites whatever the byte walk had stored.
template <int main_mode, bool BIN>
__global__ __launch_bounds__(256) void k_sobel_nms_rows(const ImgDesc* __restrict__ desc, Geo g, const uint8_t* __restrict__ planes,
                                                        uint8_t* __restrict__ maps, uint8_t* __restrict__ edges, int v_first, int low,
                                                        int high, int high_main, int* __restrict__ weak,
                                                        int* __restrict__ weak_main, const int* __restrict__ has_colour,
                                                        const int* __restrict__ band_flags, int gx, int gy)
{
    const TileId tl = tile_of_block(gx, gy);
    // image-major: every XCD's contiguous share of the grid (tile_of_block) then holds whole images with all their planes -- with the
    // planes outermost an XCD got one KIND of plane, and the byte walk's gain on the two-valued ones was hidden behind the XCDs
    // that held Gaussian planes (7.3 us per diagram either way; profiles/r04_c_canny_bin.txt)
    const int nv = main_mode ? 1 : NVAR - v_first;
    const int b = tl.z / nv;
    const int v = main_mode ? 0 : v_first + tl.z % nv;
    constexpr int NC = main_mode == 3 ? 3 : 1;                     // channels whose gradients compete
    const ImgDesc im = desc[b];
    // the main Canny of an image runs here, on its grey plane, unless the image really is coloured (k_grey's has_colour: see there)
    const bool grey_like = im.cn == 1 || (main_mode != 0 && has_colour[b] == 0);
    if (main_mode == 1 && !grey_like) return;
    if (main_mode == 3 && grey_like) return;
    const bool main_out = main_mode != 0 && (grey_like || main_mode == 3);     // writes map 0 + edges
    if (main_mode == 1) high = high_main;
    const int w = im.w, h = im.h;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int cgp = tl.tx * 4 + wave;                              // 256-pixel column group
    const int x0 = (cgp * 64 + lane) * 4;
    const int y0 = tl.ty * CR_R;
    if (cgp * 256 >= w || y0 >= h) return;
    const bool active = x0 < w;
    const uint8_t* plane = main_mode == 3 ? planes + (size_t)b * g.slot : (v == 0 ? im.grey : planes + ((size_t)v * g.nb + b) * g.slot);
    const int sp = (v == 0 && main_mode != 3) ? im.gpitch : g.pitch;
    const int m_first = (main_mode == 1 || main_mode == 3) ? 0 : 1 + v;
    uint8_t* mp = maps + ((size_t)m_first * g.nb + b) * g.slot;
    uint8_t* mp0 = (main_mode == 2 && main_out) ? maps + (size_t)b * g.slot : nullptr;
    uint8_t* ep = main_out ? edges + (size_t)b * g.slot : nullptr;
    BlBuf pbuf[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) pbuf[c] = bl_buf(plane + (size_t)c * g.nb * g.slot);
    const BlBuf mbuf = bl_buf(mp), m0buf = bl_buf(mp0 ? mp0 : mp), ebuf = bl_buf(ep ? ep : mp);
    CrThr th;
    th.lowp1 = (unsigned)(iclamp(low, -1, 4095) + 1) * 0x00010001u;
    th.highp1 = (unsigned)(iclamp(high, -1, 4095) + 1) * 0x00010001u;
    th.high0p1 = (unsigned)(iclamp(high_main, -1, 4095) + 1) * 0x00010001u;

    // BORDER_REPLICATE along x: byte k of the (L, M, R) triple is pixel x0 - 4 + k; lanes whose 3 x 4 bytes reach outside the
    // image rebuild the triple with byte permutes (selectors computed once)
    const bool fix_lane = active && (x0 - 4 < 0 || x0 + 7 >= w);
    const bool fix = __any(fix_lane ? 1 : 0) != 0;
    unsigned sL = 0x03020100u, sM = 0x07060504u, sR = 0x07060504u;      // L' = perm(M, L), M' = perm(M, L), R' = perm(R, M)
    if (fix_lane) {
        sL = 0; sM = 0; sR = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int l = iclamp(x0 - 4 + j, 0, w - 1) - (x0 - 4), m = iclamp(x0 + j, 0, w - 1) - (x0 - 4);
            const int r = iclamp(x0 + 4 + j, 0, w - 1) - x0;                       // index into (M, R)
            sL |= (unsigned)iclamp(l, 0, 7) << (8 * j);
            sM |= (unsigned)iclamp(m, 0, 7) << (8 * j);
            sR |= (unsigned)iclamp(r, 0, 7) << (8 * j);
        }
    }
    // columns that exist in the image: gradients (hence magnitudes) outside are 0
    const unsigned k01 = (x0 < w ? 0x0000ffffu : 0u) | (x0 + 1 < w ? 0xffff0000u : 0u);
    const unsigned k23 = (x0 + 2 < w ? 0x0000ffffu : 0u) | (x0 + 3 < w ? 0xffff0000u : 0u);
    // the extra magnitude of the wavefront's end lanes: column xe = x0 - 1 (lane 0) or x0 + 4 (lane 63)
    const bool e_hi = lane == 63;
    const int xe = e_hi ? x0 + 4 : x0 - 1;
    const bool e_ok = (lane == 0 || lane == 63) && xe >= 0 && xe < w;
    const unsigned eS = e_hi ? 0x0c050403u : 0x0c040302u;              // bytes (xe - 1, xe, xe + 1) of perm(hi, lo): lane 63: (M, R) -> M.b3, R.b0, R.b1 ; lane 0: (L, M) -> L.b2, L.b3, M.b0
    unsigned vm = 0;                                                    // bytes of the output dword that are pixels of the image
#pragma unroll
    for (int q = 0; q < 4; q++) if (x0 + q < w) vm |= 0xffu << (8 * q);

    const bool has_e = (lane == 0 && x0 >= 4) || (lane == 63 && x0 + 4 < w);
    const unsigned xm = active ? (unsigned)x0 : 0u, xeo = has_e ? (unsigned)(lane == 0 ? x0 - 4 : x0 + 4) : 0u;

    unsigned PA[NC][3], PB[NC][3], PC[NC][3], PE[NC][3];   // per channel: pixel pairs (-1,0), (1,2), (3,4) and the end-lane bytes of the last 3 input rows
    // magnitudes of the last 3 gradient rows: the own pairs (pixels 0,1 and 2,3) and the same row shifted by one pixel --
    // (-1,0), (1,2), (3,4), formed ONCE per row from the neighbour lanes' pairs (each row is used by three suppressions)
    unsigned M01[3], M23[3], SL[3], SM[3], SR[3];
    // of the last 2 gradient rows: |dx|, |dy| and (bit 15) "signs differ" (the suppression needs nothing else of the gradients)
    unsigned AX01[2], AX23[2], AY01[2], AY23[2], SG01[2], SG23[2];
#pragma unroll
    for (int i = 0; i < 3; i++) {
#pragma unroll
        for (int c = 0; c < NC; c++) PA[c][i] = PB[c][i] = PC[c][i] = PE[c][i] = 0;
        M01[i] = M23[i] = SL[i] = SM[i] = SR[i] = 0;
    }
#pragma unroll
    for (int i = 0; i < 2; i++) { AX01[i] = AX23[i] = AY01[i] = AY23[i] = SG01[i] = SG23[i] = 0; }

    unsigned nM[NC], nE[NC];
    {
        const int ro = rowoff(iclamp(y0 - 2, 0, h - 1), sp);
#pragma unroll
        for (int c = 0; c < NC; c++) { nM[c] = bl_bload(pbuf[c], ro, xm); nE[c] = bl_bload(pbuf[c], ro, xeo); }
    }
    unsigned wk_acc = 0, wk0_acc = 0;
    static_assert((CR_R + 4) % 6 == 0, "the row loop is unrolled by the ring depths (3 and 2)");
    const int t_end = imin(CR_R + 4, h + 2 - (y0 - 2));               // input rows beyond h + 1 feed no output of this band
    if constexpr (BIN && NC == 1) {
        // not tried: Gaussian planes (never two-valued, short of a constant image); the grey plane and the medians of a flagged band
        // (band_flags: the HoughCircles dispatch runs behind k_blur and sees its verdicts, the main Canny runs before it and sees what
        // k_grey flagged -- nothing for sources that are their own grey plane); bands whose first row already fails
        bool try_bytes = !(main_mode == 0 && v >= 3 && (v & 1));
        if (try_bytes && !(main_mode == 0 && v == 1) && band_flags != nullptr)
            try_bytes = band_flags[((size_t)b * mb_bands_y(g.hmax) + (y0 / MB_R)) * mb_bands_x(g.wmax) + cgp] == 0;
        unsigned ve = 0;                                                // pixels of the end-lane dword that exist (the two-valued test looks at them too)
        if (has_e) {
#pragma unroll
            for (int q = 0; q < 4; q++) if (lane == 0 || x0 + 4 + q < w) ve |= 0xffu << (8 * q);
        }
        const unsigned vmt = active ? vm : 0u;
        if (try_bytes && __any(((((nM[0] >> 1) ^ nM[0]) & vmt) | (((nE[0] >> 1) ^ nE[0]) & ve)) & 0x7f7f7f7fu ? 1 : 0)) try_bytes = false;
        if (try_bytes) {
            constexpr unsigned ONES = 0x01010101u, TOP = 0x80808080u, FOUR = 0x04040404u;
            // thresholds in units of 255: magnitude m (units) passes  255 m > thr  <=>  m >= floor(thr / 255) + 1
            const unsigned tlb = (unsigned)(low < 0 ? 0 : imin(low / 255 + 1, 100)) * ONES;
            const unsigned thb = (unsigned)(high < 0 ? 0 : imin(high / 255 + 1, 100)) * ONES;
            const unsigned th0b = (unsigned)(high_main < 0 ? 0 : imin(high_main / 255 + 1, 100)) * ONES;
            unsigned HD[3], RS[3], BE[3];     // of the last 3 input rows: right - left + 1, left + 2 centre + right, the end-lane bytes
            unsigned MG[3], SLb[3], SRb[3];   // magnitudes of the last 3 gradient rows: own pixels, shifted by one pixel to either side
            unsigned S22[2], S67[2], SGf[2];  // of the last 2 gradient rows, in bit 7 of each byte: the two axis sectors, "signs differ"
#pragma unroll
            for (int i = 0; i < 3; i++) HD[i] = RS[i] = BE[i] = MG[i] = SLb[i] = SRb[i] = 0;
#pragma unroll
            for (int i = 0; i < 2; i++) S22[i] = S67[i] = SGf[i] = 0;
            unsigned odd = 0;
            bool two_valued = true;
            // the byte walk spends too few cycles on a row for ONE row of loads in flight to cover the memory latency: three rows ahead,
            // a statically indexed ring; every store on every path (BL_NO_STORE) so that the compiler's vmcnt counts what is really pending
            unsigned rM[3], rE[3];
            rM[0] = nM[0]; rE[0] = nE[0];
#pragma unroll
            for (int i = 1; i < 3; i++) {
                const int ro = rowoff(iclamp(y0 - 2 + i, 0, h - 1), sp);
                rM[i] = bl_bload(pbuf[0], ro, xm); rE[i] = bl_bload(pbuf[0], ro, xeo);
            }
            for (int t0 = 0; t0 < t_end; t0 += 6) {
                if (__any((odd & 0x7f7f7f7fu) != 0u ? 1 : 0)) { two_valued = false; break; }
#pragma unroll
                for (int u = 0; u < 6; u++) {
                    const int t = t0 + u;
                    const int yi = y0 - 2 + t;
                    const unsigned M = rM[u % 3], E = rE[u % 3];
                    {
                        const int ro = rowoff(iclamp(yi + 3, 0, h - 1), sp);
                        rM[u % 3] = bl_bload(pbuf[0], ro, xm); rE[u % 3] = bl_bload(pbuf[0], ro, xeo);
                    }
                    odd |= (((M >> 1) ^ M) & vmt) | (((E >> 1) ^ E) & ve);      // (& 0x7f7f7f7f at the test)
                    const int ps = u % 3;
                    unsigned L = bl_from_prev_lane(M, E), R = bl_from_next_lane(M, E), Mf = M;
                    if (fix) {
                        BL_KEEP_BRANCH();
                        const unsigned l2 = __builtin_amdgcn_perm(M, L, sL), m2 = __builtin_amdgcn_perm(M, L, sM), r2 = __builtin_amdgcn_perm(R, M, sR);
                        L = l2; Mf = m2; R = r2;
                    }
                    BE[ps] = __builtin_amdgcn_perm(e_hi ? R : Mf, e_hi ? Mf : L, eS) & 0x00010101u;
                    {
                        const unsigned bl = L & ONES, bm = Mf & ONES, br = R & ONES;
                        const unsigned lf = alignbyte(bm, bl, 3), rt = alignbyte(br, bm, 1);       // pixels x - 1 and x + 1 of the lane's four
                        RS[ps] = lf + rt + bm + bm;                    // (v_lshlrev is a half-rate instruction, v_add is not)
                        HD[ps] = rt + ONES - lf;
                    }
                    const int yg = yi - 1;
                    const int gs = u % 3, g2 = u % 2;
                    if (t >= 2) {
                        const int top = (u + 1) % 3, mid = (u + 2) % 3, bot = ps;
                        const unsigned inm = (yg >= 0 && yg < h) ? vm : 0u;             // pixels whose gradient exists
                        // dx + 4 and dy + 4 (0 .. 8); zero gradients outside the image
                        const unsigned dxb = bitop3<0xCA>(inm, HD[top] + HD[mid] + HD[mid] + HD[bot], FOUR);
                        const unsigned dyb = bitop3<0xCA>(inm, RS[bot] + FOUR - RS[top], FOUR);
                        // |d| = |i - 4| as a table of i & 7 (i = 8 and i = 0 share an entry: both are 4)
                        const unsigned ax = __builtin_amdgcn_perm(0x03020100u, 0x01020304u, dxb & 0x07070707u);
                        const unsigned ay = __builtin_amdgcn_perm(0x03020100u, 0x01020304u, dyb & 0x07070707u);
                        const unsigned mag = ax + ay;
                        SGf[g2] = ((dxb | TOP) - FOUR) ^ ((dyb | TOP) - FOUR);          // bit 7 of (d + 4 | 0x80) - 4: d >= 0 (0 counts as positive: OpenCV's (dx ^ dy) < 0)
                        // |dy| 2^15 < |dx| 13573  <=>  |dy| < (0, 1, 1, 2, 2)[|dx|];  |dy| 2^15 > |dx| 79109  <=>  |dy| >= (1, 3, -, -, -)[|dx|]
                        const unsigned t22 = __builtin_amdgcn_perm(0x00000002u, 0x02010100u, ax);
                        const unsigned t67 = __builtin_amdgcn_perm(0x00000009u, 0x09090301u, ax);
                        const unsigned ayh = ay | TOP;
                        S22[g2] = ~(ayh - t22); S67[g2] = ayh - t67;
                        // end lanes: bytes (xe - 1, xe, xe + 1) of the three rows are 0 / 1, signed dot products do
                        const unsigned et = BE[top], em = BE[mid], eb = BE[bot];
                        const int edx = __builtin_amdgcn_sdot4((int)et, 0x000100ff, __builtin_amdgcn_sdot4((int)em, 0x000200fe, __builtin_amdgcn_sdot4((int)eb, 0x000100ff, 0, false), false), false);
                        const int edy = __builtin_amdgcn_sdot4((int)eb, 0x00010201, __builtin_amdgcn_sdot4((int)et, 0x00fffeff, 0, false), false);
                        const unsigned mge = (e_ok && yg >= 0 && yg < h) ? (unsigned)(iabs_(edx) + iabs_(edy)) : 0u;
                        MG[gs] = mag;
                        const unsigned ml = bl_from_prev_lane(mag, mge << 24), mr = bl_from_next_lane(mag, mge);
                        SLb[gs] = alignbyte(mag, ml, 3);
                        SRb[gs] = alignbyte(mr, mag, 1);
                    }
                    const int yn = yi - 2;
                    unsigned outw = ONES, outw0 = ONES;
                    const bool emit = t >= 4 && yn < h;
                    if (emit) {
                        const int ra = (u + 1) % 3, rc = (u + 2) % 3, rb = gs, gq = g2 ^ 1;
                        const unsigned curh = MG[rc] | TOP;
                        const unsigned ktl = curh - tlb;
                        if (__any((ktl & TOP) != 0u ? 1 : 0)) {
                            // every comparison leaves its answer in bit 7 of the byte ((a | 0x80) - b: a >= b), and the sector decides which pair counts:
                            // no byte masks, no selects.  OpenCV keeps m > first && m >= second on the axes, m > both on a diagonal.
                            const unsigned cur1 = curh - ONES;                                   // (.. - b): a > b
                            const unsigned kh = bitop3<0x80>(S22[gq], cur1 - SLb[rc], curh - SRb[rc]);
                            const unsigned kv = bitop3<0x80>(S67[gq], cur1 - MG[ra], curh - MG[rb]);
                            const unsigned kd = bitop3<0xCA>(SGf[gq], (cur1 - SRb[ra]) & (cur1 - SLb[rb]), (cur1 - SLb[ra]) & (cur1 - SRb[rb]));
                            const unsigned k7 = bitop3<0xFE>(kh, kv, bitop3<0x02>(S22[gq], S67[gq], kd)) & ktl;       // bit 7: kept
                            outw = (((k7 ^ TOP) >> 7) & ONES) | (((k7 & (curh - thb)) >> 6) & 0x02020202u);
                            if (main_mode == 2) outw0 = (((k7 ^ TOP) >> 7) & ONES) | (((k7 & (curh - th0b)) >> 6) & 0x02020202u);
                        }
                        outw = (outw & vm) | (ONES & ~vm);
                        wk_acc |= (outw - ONES) & ~outw & TOP;
                        if (main_mode == 2) {
                            outw0 = (outw0 & vm) | (ONES & ~vm);
                            wk0_acc |= (outw0 - ONES) & ~outw0 & TOP;
                        }
                    }
                    {
                        const int off = rowoff(emit ? yn : 0, g.pitch);
                        const unsigned xs = (emit && active) ? xm : BL_NO_STORE;
                        bl_bstore(mbuf, off, xs, outw);
                        if (main_mode == 2) bl_bstore(m0buf, off, mp0 ? xs : BL_NO_STORE, outw0);
                        if (main_mode != 0) { const unsigned outm = mp0 ? outw0 : outw; bl_bstore(ebuf, off, ep ? xs : BL_NO_STORE, ((outm >> 1) & ONES) * 0xffu); }
                    }
                }
            }
            if (two_valued && __any((odd & 0x7f7f7f7fu) != 0u ? 1 : 0)) two_valued = false;
            if (two_valued) {
                // the band is one row of 64 x 32 hysteresis tiles: one key per tile (16 lanes) that holds a weak pixel
                const unsigned long long bw = __ballot(wk_acc != 0u && active), bw0 = __ballot(mp0 != nullptr && wk0_acc != 0u && active);
                if ((lane & 15) == 0) {
                    const int tile_x = cgp * 4 + (lane >> 4), tile_y = y0 / CT_H;
                    const unsigned long long grp = 0xffffull << lane;
                    int* weak_first = main_mode == 1 ? weak_main : weak;
                    if (bw & grp) weak_first[1 + atomicAdd(&weak_first[0], 1)] = (int)(((size_t)m_first * g.nb + b) * g.tiles + (size_t)tile_y * g.tw + tile_x);
                    if (bw0 & grp) weak_main[1 + atomicAdd(&weak_main[0], 1)] = (int)((size_t)b * g.tiles + (size_t)tile_y * g.tw + tile_x);
                }
                return;
            }
            wk_acc = 0; wk0_acc = 0;
            const int ro = rowoff(iclamp(y0 - 2, 0, h - 1), sp);       // start over (the 16-bit walk below)
            nM[0] = bl_bload(pbuf[0], ro, xm); nE[0] = bl_bload(pbuf[0], ro, xeo);
        }
    }
    // single-channel walks keep three rows of loads in flight and issue every store on every path, like the byte walk (noisy diagrams:
    // main Canny 1.69 -> 1.62 us, the seven Cannys 6.95 -> 6.85); the colour mode (6 loads per row) waits for the next row before it stores
    constexpr bool DEEP = NC == 1;
    unsigned rM[3], rE[3];
    if (DEEP) {
        rM[0] = nM[0]; rE[0] = nE[0];
#pragma unroll
        for (int i = 1; i < 3; i++) {
            const int ro = rowoff(iclamp(y0 - 2 + i, 0, h - 1), sp);
            rM[i] = bl_bload(pbuf[0], ro, xm); rE[i] = bl_bload(pbuf[0], ro, xeo);
        }
    }
    for (int t0 = 0; t0 < t_end; t0 += 6) {
#pragma unroll
        for (int u = 0; u < 6; u++) {
            const int t = t0 + u;
            const int yi = y0 - 2 + t;                                 // input row (clamped when outside)
            unsigned Mc[NC], Ec[NC];
            if (DEEP) {
                Mc[0] = rM[u % 3]; Ec[0] = rE[u % 3];
                const int ro = rowoff(iclamp(yi + 3, 0, h - 1), sp);
                rM[u % 3] = bl_bload(pbuf[0], ro, xm); rE[u % 3] = bl_bload(pbuf[0], ro, xeo);
            } else {
#pragma unroll
            for (int c = 0; c < NC; c++) { Mc[c] = nM[c]; Ec[c] = nE[c]; }
            {
                const int ro = rowoff(iclamp(yi + 1, 0, h - 1), sp);
#pragma unroll
                for (int c = 0; c < NC; c++) { nM[c] = bl_bload(pbuf[c], ro, xm); nE[c] = bl_bload(pbuf[c], ro, xeo); }
            }
            }
            const int ps = u % 3;                                      // ring slot of this input row
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const unsigned M = Mc[c], E = Ec[c];
                unsigned L = bl_from_prev_lane(M, E), R = bl_from_next_lane(M, E), Mf = M;
                if (fix) {
                    BL_KEEP_BRANCH();
                    const unsigned l2 = __builtin_amdgcn_perm(M, L, sL), m2 = __builtin_amdgcn_perm(M, L, sM), r2 = __builtin_amdgcn_perm(R, M, sR);
                    L = l2; Mf = m2; R = r2;
                }
                PA[c][ps] = __builtin_amdgcn_perm(Mf, L, 0x0c040c03u);
                PB[c][ps] = __builtin_amdgcn_perm(Mf, Mf, 0x0c020c01u);
                PC[c][ps] = __builtin_amdgcn_perm(R, Mf, 0x0c040c03u);
                PE[c][ps] = __builtin_amdgcn_perm(e_hi ? R : Mf, e_hi ? Mf : L, eS);
            }
            // gradient row yg = yi - 1 from input rows yi - 2, yi - 1, yi (slots ps + 1, ps + 2, ps)
            const int yg = yi - 1;
            const int gs = u % 3;                                      // magnitude ring slot of gradient row yg (same phase as ps)
            const int g2 = u % 2;                                      // gradient ring slot
            if (t >= 2) {
                const int top = (u + 1) % 3, mid = (u + 2) % 3, bot = ps;
                unsigned dx01 = 0, dx23 = 0, dy01 = 0, dy23 = 0, mg01 = 0, mg23 = 0, mge = 0;
                // gradient rows outside the image (yg = -1, h) are zero: folded into the column masks (a branch around the block
                // costs seven register clears on every row for the sake of two rows per image)
                const unsigned rowm = (yg >= 0 && yg < h) ? 0xffffffffu : 0u;
#pragma unroll
                for (int c = 0; c < NC; c++) {
                    const v2s ta = pk_from(PA[c][top]), tb = pk_from(PB[c][top]), tc = pk_from(PC[c][top]);
                    const v2s ma = pk_from(PA[c][mid]), mb = pk_from(PB[c][mid]), mc = pk_from(PC[c][mid]);
                    const v2s ba = pk_from(PA[c][bot]), bb = pk_from(PB[c][bot]), bc = pk_from(PC[c][bot]);
                    const v2s ca = ta + ma + ma + ba, cb = tb + mb + mb + bb, cc = tc + mc + mc + bc;
                    const v2s da = ba - ta, db = bb - tb, dc = bc - tc;
                    const v2s x01 = cb - ca, x23 = cc - cb;
                    const v2s m01 = pk_from(__builtin_amdgcn_alignbit(pk_bits(db), pk_bits(da), 16));    // (dif0, dif1)
                    const v2s m23 = pk_from(__builtin_amdgcn_alignbit(pk_bits(dc), pk_bits(db), 16));    // (dif2, dif3)
                    const v2s y01 = da + m01 + m01 + db, y23 = db + m23 + m23 + dc;
                    const unsigned r01 = k01 & rowm, r23 = k23 & rowm;
                    const unsigned cdx01 = pk_bits(x01) & r01, cdy01 = pk_bits(y01) & r01;
                    const unsigned cdx23 = pk_bits(x23) & r23, cdy23 = pk_bits(y23) & r23;
                    const unsigned cm01 = pk_bits(pk_abs(pk_from(cdx01)) + pk_abs(pk_from(cdy01)));
                    const unsigned cm23 = pk_bits(pk_abs(pk_from(cdx23)) + pk_abs(pk_from(cdy23)));
                    // end lanes: magnitude at column xe from bytes (xe - 1, xe, xe + 1) of the three rows
                    const unsigned et = PE[c][top], em = PE[c][mid], eb = PE[c][bot];
                    const int sc = (int)__builtin_amdgcn_udot4(et, 0x00010000u, __builtin_amdgcn_udot4(em, 0x00020000u, __builtin_amdgcn_udot4(eb, 0x00010000u, 0u, false), false), false);
                    const int sa = (int)__builtin_amdgcn_udot4(et, 0x00000001u, __builtin_amdgcn_udot4(em, 0x00000002u, __builtin_amdgcn_udot4(eb, 0x00000001u, 0u, false), false), false);
                    const int edy = (int)__builtin_amdgcn_udot4(eb, 0x00010201u, 0u, false) - (int)__builtin_amdgcn_udot4(et, 0x00010201u, 0u, false);
                    const unsigned cme = e_ok ? (unsigned)(iabs_(sc - sa) + iabs_(edy)) & rowm : 0u;
                    if (c == 0) { dx01 = cdx01; dy01 = cdy01; dx23 = cdx23; dy23 = cdy23; mg01 = cm01; mg23 = cm23; mge = cme; }
                    else {
                        // a later channel wins only with a strictly larger magnitude (per pixel: the 16-bit halves decide separately)
                        const unsigned w01 = pk_gt(cm01, mg01), w23 = pk_gt(cm23, mg23);
                        dx01 = bsel(w01, cdx01, dx01); dy01 = bsel(w01, cdy01, dy01); mg01 = bsel(w01, cm01, mg01);
                        dx23 = bsel(w23, cdx23, dx23); dy23 = bsel(w23, cdy23, dy23); mg23 = bsel(w23, cm23, mg23);
                        mge = cme > mge ? cme : mge;
                    }
                }
                {
                    const unsigned ax01 = pk_bits(pk_abs(pk_from(dx01))), ay01 = pk_bits(pk_abs(pk_from(dy01)));
                    const unsigned ax23 = pk_bits(pk_abs(pk_from(dx23))), ay23 = pk_bits(pk_abs(pk_from(dy23)));
                    AX01[g2] = ax01; AX23[g2] = ax23; AY01[g2] = ay01; AY23[g2] = ay23;
                    SG01[g2] = dx01 ^ dy01; SG23[g2] = dx23 ^ dy23;               // bit 15 of each half: the signs differ
                }
                M01[gs] = mg01; M23[gs] = mg23;
                // pair (x0 - 2, x0 - 1) of the left neighbour and pair (x0 + 4, x0 + 5) of the right one (only their inner halves are used)
                const unsigned ml = bl_from_prev_lane(mg23, mge << 16), mr = bl_from_next_lane(mg01, mge);
                SL[gs] = __builtin_amdgcn_alignbit(mg01, ml, 16);
                SM[gs] = __builtin_amdgcn_alignbit(mg23, mg01, 16);
                SR[gs] = __builtin_amdgcn_alignbit(mr, mg23, 16);
            }
            // suppression of row yn = yg - 1 (magnitude rows yn - 1, yn, yn + 1 = slots gs + 1, gs + 2, gs; gradients in slot g2 ^ 1)
            const int yn = yi - 2;
            unsigned outw = 0x01010101u, outw0 = 0x01010101u;
            const bool emit = t >= 4 && yn < h;
            if (emit) {
                const int ra = (u + 1) % 3, rc = (u + 2) % 3, rb = gs, gq = g2 ^ 1;
                const unsigned mb01 = M01[rc], mb23 = M23[rc];
                const int mxall = imax(imax((int)(mb01 & 0xffffu), (int)(mb01 >> 16)), imax((int)(mb23 & 0xffffu), (int)(mb23 >> 16)));
                if (__any(mxall > low ? 1 : 0)) {
                    unsigned o0, o0m = 0, o1, o1m = 0;
                    constexpr bool TWO = main_mode == 2;
                    cr_nms_pair<TWO>(M01[rc], SL[rc], SM[rc], M01[ra], M01[rb], SL[ra], SM[ra], SL[rb], SM[rb], AX01[gq], AY01[gq], SG01[gq], th, o0, o0m);
                    cr_nms_pair<TWO>(M23[rc], SM[rc], SR[rc], M23[ra], M23[rb], SM[ra], SR[ra], SM[rb], SR[rb], AX23[gq], AY23[gq], SG23[gq], th, o1, o1m);
                    outw = __builtin_amdgcn_perm(o1, o0, 0x06040200u);
                    if (TWO) outw0 = __builtin_amdgcn_perm(o1m, o0m, 0x06040200u);
                }
                outw = (outw & vm) | (0x01010101u & ~vm);
                wk_acc |= (outw - 0x01010101u) & ~outw & 0x80808080u;            // some byte == 0
                if (main_mode == 2) {
                    outw0 = (outw0 & vm) | (0x01010101u & ~vm);
                    wk0_acc |= (outw0 - 0x01010101u) & ~outw0 & 0x80808080u;
                }
            }
            // stores after the wait for the prefetched row (see BL_CONSUME in k_filters.h)
            if (DEEP) {
                const int off = rowoff(emit ? yn : 0, g.pitch);
                const unsigned xs = (emit && active) ? xm : BL_NO_STORE;
                bl_bstore(mbuf, off, xs, outw);
                if (main_mode == 2) bl_bstore(m0buf, off, mp0 ? xs : BL_NO_STORE, outw0);
                if (main_mode != 0) { const unsigned outm = mp0 ? outw0 : outw; bl_bstore(ebuf, off, ep ? xs : BL_NO_STORE, ((outm >> 1) & 0x01010101u) * 0xffu); }
            } else {
            BL_SCHED_FENCE();
#pragma unroll
            for (int c = 0; c < NC; c++) BL_CONSUME(nM[c], nE[c]);
            BL_SCHED_FENCE();
            if (emit && active) {
                const int off = rowoff(yn, g.pitch);
                bl_bstore(mbuf, off, xm, outw);
                if (mp0) bl_bstore(m0buf, off, xm, outw0);
                if (ep) { const unsigned outm = mp0 ? outw0 : outw; bl_bstore(ebuf, off, xm, ((outm >> 1) & 0x01010101u) * 0xffu); }
            }
            }
            // hysteresis worklist: at the last row of a row of 64 x 32 tiles, one key per tile (16 lanes) that holds a weak pixel
            if (emit && ((yn & (CT_H - 1)) == CT_H - 1 || yn == h - 1)) {
                const unsigned long long bw = __ballot(wk_acc != 0u && active), bw0 = __ballot(mp0 != nullptr && wk0_acc != 0u && active);
                if ((lane & 15) == 0) {
                    const int tile_x = cgp * 4 + (lane >> 4), tile_y = yn / CT_H;
                    const unsigned long long grp = 0xffffull << lane;
                    int* weak_first = (main_mode == 1 || main_mode == 3) ? weak_main : weak;
                    if (bw & grp) weak_first[1 + atomicAdd(&weak_first[0], 1)] = (int)(((size_t)m_first * g.nb + b) * g.tiles + (size_t)tile_y * g.tw + tile_x);
                    if (bw0 & grp) weak_main[1 + atomicAdd(&weak_main[0], 1)] = (int)((size_t)b * g.tiles + (size_t)tile_y * g.tw + tile_x);
                }
                wk_acc = 0; wk0_acc = 0;
            }
        }
    }
}

}  // namespace i2s
