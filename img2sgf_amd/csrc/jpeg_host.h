// JPEG front end: marker parsing, and the serial form of the Huffman entropy decoding -- one bit stream per
// scan -- into quantised DCT coefficient blocks.  Everything after that (dequantisation, inverse DCT, chroma upsampling,
// colour conversion: csrc/k_jpeg.h) runs on the device, and so does the entropy decoding of sequential files
// (k_jpeg_entropy.h).  The reference reaches the same pixels through PIL.Image.open(path).convert("RGB") (img2sgf.py:651), i.e. libjpeg-turbo with its defaults (JDCT_ISLOW, fancy upsampling);
// this file and k_jpeg.h restate exactly that decoder for 8-bit Huffman JPEGs, sequential (SOF0 / SOF1) and progressive (SOF2,
// spectral selection + successive approximation), 1 or 3 components, 4:4:4 / 4:2:2 / 4:2:0, any scan script that ends with
// coefficients 0..9 of every component fully refined (Al = 0): Pillow leaves libjpeg's do_block_smoothing on, and
// smoothing_ok() (jdcoefct.c) switches decompress_smooth_data on whenever one of those coefficients is not -- a DC-only
// script, a final Al > 0, a file cut between scans; such files are reported as unsupported, like everything else that is not
// restated here (arithmetic coding, lossless, CMYK, RGB-coded, 12-bit, a file that ends without EOI -- Pillow decides what a
// truncated file means), so that the caller can decode it elsewhere; nothing is approximated.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include <vector>

// The serial entropy decoder below (bit reader, Huffman symbols, sequential and progressive block decoding) is written once and
// compiled twice: for the host (one file per host thread: progressive files, and every file when i2s_params.jpeg_entropy_device
// is 0) and for the device (k_jpeg_huffman, k_jpeg.h: one file per lane; jpeg_entropy_device 2).  Sequential files are normally
// decoded by k_jpeg_entropy.h, parallel inside each scan; this decoder is its yardstick.  The marker parser stays host-only.
#if defined(__HIPCC__) || defined(HIPEMU)
#define I2S_HD __host__ __device__
#else
#define I2S_HD
#endif

namespace i2s {

struct JpegComp {
    int id, h, v, tq, td, ta;
    int bw, bh;          // blocks per row / column (padded to whole MCUs)
    int dw, dh;          // real (downsampled) sample dimensions
};

struct JpegHuff {
    bool present = false;
    uint8_t look_len[512];     // code length for the 9-bit prefix (0 = longer than 9 bits)
    uint8_t look_sym[512];
    int maxcode[18];           // largest code of each length (-1 if none), as in jdhuff.c
    int valoff[17];
    uint8_t syms[256];
};

struct JpegRst { size_t pos, nstuff; };      // an RSTn marker: offset of its FF inside the scan's data, stuffed FF00 pairs before it
struct JpegScan {
    int ns = 0, ci[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    int ss = 0, se = 63, ah = 0, al = 0, dri = 0;
    JpegHuff dc[4], ac[4];     // the tables in force at this SOS (they may be redefined between scans)
    const uint8_t* data = nullptr;
    size_t len = 0;
    // what the walk over the entropy-coded data met (the parallel decoder lays its segments out from this)
    size_t nstuff = 0;         // stuffed FF00 pairs
    bool clean = true;         // nothing but stuffed bytes and RSTn markers after an FF (no fill bytes, no FF as the last byte)
    std::vector<JpegRst> rst;
};

struct JpegFile {
    int X = 0, Y = 0, ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    bool progressive = false;
    JpegComp c[3];
    uint16_t q[4][64];         // natural (row-major) order, as in force when the FIRST scan of a component starts (libjpeg latches them there)
    uint16_t qc[3][64];        // per component
    bool have_q[4] = {false, false, false, false};
    std::vector<JpegScan> scans;
};

enum { JPG_OK = 0, JPG_BAD = 1, JPG_UNSUPPORTED = 2,
       JPG_REDO = 3 };   // device verdict only: decode this file with the serial decoder instead (jpeg_entropy_pass)

static const uint8_t JPG_ZZ[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                   41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                   30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ const uint8_t JPG_ZZ_DEV[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                                        41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                                        30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};
#define JPG_ZZ_AT(k) JPG_ZZ_DEV[k]
#else
#define JPG_ZZ_AT(k) JPG_ZZ[k]
#endif

static int jpg_build_huff(const uint8_t* counts, const uint8_t* syms, int nsyms, JpegHuff* h)
{
    memset(static_cast<void*>(h), 0, sizeof(*h));      // padding and the unused entries too: equal tables compare equal byte for byte (the passes pool them)
    memcpy(h->syms, syms, (size_t)nsyms);
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
        h->valoff[l] = k - code;
        for (int i = 0; i < counts[l - 1]; i++, k++, code++) {
            if (l <= 9) {
                const int lo = code << (9 - l), n = 1 << (9 - l);
                if (lo + n > 512) return JPG_BAD;
                for (int j = 0; j < n; j++) { h->look_len[lo + j] = (uint8_t)l; h->look_sym[lo + j] = syms[k]; }
            }
        }
        h->maxcode[l] = counts[l - 1] ? code - 1 : -1;
        if (code > (1 << l)) return JPG_BAD;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
    h->present = true;
    return JPG_OK;
}

// Parses the whole file: frame header, tables, and every scan (with the entropy-coded segment that follows it).
static int jpg_parse(const uint8_t* d, size_t n, JpegFile* f)
{
    if (n < 4 || d[0] != 0xFF || d[1] != 0xD8) return JPG_BAD;
    size_t p = 2;
    bool have_frame = false, adobe_rgb = false, latched[3] = {false, false, false};
    JpegHuff dc[4], ac[4];
    memset(static_cast<void*>(dc), 0, sizeof(dc)); memset(static_cast<void*>(ac), 0, sizeof(ac));
    int dri = 0;
    int cbits[3][10];                 // libjpeg's coef_bits for coefficients 0..9 (zigzag order): -1 = never sent, else the last Al
    for (int c = 0; c < 3; c++) for (int k = 0; k < 10; k++) cbits[c][k] = -1;
    for (;;) {
        // the data end without an EOI marker (or in the middle of one): what Pillow makes of such a file depends on where it
        // was cut, so it is not decoded here
        if (p + 2 > n || d[p] != 0xFF) return f->scans.empty() ? JPG_BAD : JPG_UNSUPPORTED;
        while (p + 1 < n && d[p + 1] == 0xFF) p++;                   // fill bytes
        if (p + 1 >= n) return f->scans.empty() ? JPG_BAD : JPG_UNSUPPORTED;
        const int m = d[p + 1];
        p += 2;
        if (m == 0xD9) break;
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;         // parameterless markers
        if (p + 2 > n) return JPG_BAD;
        const size_t L = ((size_t)d[p] << 8) | d[p + 1];
        if (L < 2 || p + L > n) return JPG_BAD;
        const uint8_t* s = d + p + 2;
        const size_t sl = L - 2;
        if (m == 0xDB) {
            size_t i = 0;
            while (i < sl) {
                const int pq = s[i] >> 4, tq = s[i] & 15;
                i++;
                if (tq > 3 || pq > 1) return JPG_BAD;
                if (pq == 1) return JPG_UNSUPPORTED;                 // 16-bit tables belong to 12-bit data
                if (i + 64 > sl) return JPG_BAD;
                for (int k = 0; k < 64; k++) f->q[tq][JPG_ZZ[k]] = s[i + k];
                f->have_q[tq] = true;
                i += 64;
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (sl < 6 || have_frame) return JPG_BAD;
            if (s[0] != 8) return JPG_UNSUPPORTED;
            f->progressive = m == 0xC2;
            f->Y = (s[1] << 8) | s[2]; f->X = (s[3] << 8) | s[4]; f->ncomp = s[5];
            if (f->X < 1 || f->Y < 1) return JPG_UNSUPPORTED;        // Y == 0 (DNL) is not handled
            if (f->ncomp != 1 && f->ncomp != 3) return JPG_UNSUPPORTED;
            if (sl < (size_t)(6 + 3 * f->ncomp)) return JPG_BAD;
            for (int c = 0; c < f->ncomp; c++) {
                JpegComp& jc = f->c[c];
                jc.id = s[6 + 3 * c]; jc.h = s[7 + 3 * c] >> 4; jc.v = s[7 + 3 * c] & 15; jc.tq = s[8 + 3 * c];
                if (jc.h < 1 || jc.v < 1 || jc.h > 4 || jc.v > 4 || jc.tq > 3) return JPG_BAD;
            }
            if (f->ncomp == 3) {
                // luma at full resolution, both chroma planes alike: 4:4:4, 4:2:2 (h2v1) or 4:2:0 (h2v2)
                const int h0 = f->c[0].h, v0 = f->c[0].v;
                if (f->c[1].h != 1 || f->c[1].v != 1 || f->c[2].h != 1 || f->c[2].v != 1) return JPG_UNSUPPORTED;
                if (!((h0 == 1 && v0 == 1) || (h0 == 2 && v0 == 1) || (h0 == 2 && v0 == 2))) return JPG_UNSUPPORTED;
                if (f->c[0].id == 'R' && f->c[1].id == 'G' && f->c[2].id == 'B') return JPG_UNSUPPORTED;
            } else {
                f->c[0].h = f->c[0].v = 1;     // a lone component: the sampling factors only scale the MCU, libjpeg ignores them
            }
            f->hmax = f->vmax = 1;
            for (int c = 0; c < f->ncomp; c++) { if (f->c[c].h > f->hmax) f->hmax = f->c[c].h; if (f->c[c].v > f->vmax) f->vmax = f->c[c].v; }
            f->mcux = (f->X + 8 * f->hmax - 1) / (8 * f->hmax);
            f->mcuy = (f->Y + 8 * f->vmax - 1) / (8 * f->vmax);
            for (int c = 0; c < f->ncomp; c++) {
                JpegComp& jc = f->c[c];
                jc.bw = f->mcux * jc.h; jc.bh = f->mcuy * jc.v;
                jc.dw = (f->X * jc.h + f->hmax - 1) / f->hmax; jc.dh = (f->Y * jc.v + f->vmax - 1) / f->vmax;
            }
            have_frame = true;
        } else if ((m >= 0xC3 && m <= 0xCF) && m != 0xC4 && m != 0xC8) {
            return JPG_UNSUPPORTED;                                  // lossless, hierarchical, arithmetic coding
        } else if (m == 0xC4) {
            size_t i = 0;
            while (i < sl) {
                const int tc = s[i] >> 4, th = s[i] & 15;
                i++;
                if (tc > 1 || th > 3 || i + 16 > sl) return JPG_BAD;
                int cnt = 0;
                for (int k = 0; k < 16; k++) cnt += s[i + k];
                if (cnt > 256 || i + 16 + (size_t)cnt > sl) return JPG_BAD;
                if (jpg_build_huff(s + i, s + i + 16, cnt, tc ? &ac[th] : &dc[th])) return JPG_BAD;
                i += 16 + (size_t)cnt;
            }
        } else if (m == 0xDD) {
            if (sl < 2) return JPG_BAD;
            dri = (s[0] << 8) | s[1];
        } else if (m == 0xEE) {
            // Adobe marker: transform 0 with three components means the data are RGB, not YCbCr
            if (sl >= 12 && memcmp(s, "Adobe", 5) == 0 && s[11] == 0) adobe_rgb = true;
        } else if (m == 0xDA) {
            if (!have_frame || sl < 1) return JPG_BAD;
            if (adobe_rgb && f->ncomp == 3) return JPG_UNSUPPORTED;
            JpegScan sc;
            sc.ns = s[0];
            if (sc.ns < 1 || sc.ns > f->ncomp || sl < (size_t)(1 + 2 * sc.ns + 3)) return JPG_BAD;
            for (int k = 0; k < sc.ns; k++) {
                int ci = -1;
                for (int c = 0; c < f->ncomp; c++) if (f->c[c].id == s[1 + 2 * k]) ci = c;
                if (ci < 0 || (k > 0 && ci <= sc.ci[k - 1])) return JPG_BAD;
                sc.ci[k] = ci; sc.td[k] = s[2 + 2 * k] >> 4; sc.ta[k] = s[2 + 2 * k] & 15;
                if (sc.td[k] > 3 || sc.ta[k] > 3) return JPG_BAD;
                if (!latched[ci]) {                                  // libjpeg's latch_quant_tables
                    if (!f->have_q[f->c[ci].tq]) return JPG_BAD;
                    memcpy(f->qc[ci], f->q[f->c[ci].tq], sizeof(f->qc[ci]));
                    latched[ci] = true;
                }
            }
            sc.ss = s[1 + 2 * sc.ns]; sc.se = s[2 + 2 * sc.ns]; sc.ah = s[3 + 2 * sc.ns] >> 4; sc.al = s[3 + 2 * sc.ns] & 15;
            if (f->progressive) {
                if (sc.ss > sc.se || sc.se > 63 || sc.al > 13 || sc.ah > 13) return JPG_BAD;
                if (sc.ss == 0 ? sc.se != 0 : sc.ns != 1) return JPG_BAD;      // DC scans hold DC only; AC scans one component
                if (sc.ah != 0 && sc.ah != sc.al + 1) return JPG_BAD;
            } else if (sc.ss != 0 || sc.se != 63 || sc.ah != 0 || sc.al != 0) return JPG_BAD;
            for (int k = 0; k < sc.ns; k++) {
                const bool need_dc = sc.ss == 0 && sc.ah == 0, need_ac = sc.se > 0;
                if ((need_dc && !dc[sc.td[k]].present) || (need_ac && !ac[sc.ta[k]].present)) return JPG_BAD;
            }
            for (int k = 0; k < sc.ns; k++)
                for (int z = sc.ss; z <= sc.se && z < 10; z++) cbits[sc.ci[k]][z] = sc.al;
            for (int t = 0; t < 4; t++) { sc.dc[t] = dc[t]; sc.ac[t] = ac[t]; }
            sc.dri = dri;
            // the entropy-coded segment runs up to the next marker that is neither a stuffed FF00 nor RSTn
            size_t q0 = p + L, q = q0;
            while (q < n) {
                const uint8_t* ff = static_cast<const uint8_t*>(memchr(d + q, 0xFF, n - q));
                if (!ff) { q = n; break; }
                q = (size_t)(ff - d);
                if (q + 1 >= n) { sc.clean = false; q = n; break; }
                const int nx = d[q + 1];
                if (nx == 0) { sc.nstuff++; q += 2; }
                else if (nx >= 0xD0 && nx <= 0xD7) { sc.rst.push_back(JpegRst{q - q0, sc.nstuff}); q += 2; }
                else if (nx == 0xFF) { sc.clean = false; q++; }      // fill byte
                else break;
            }
            sc.data = d + q0; sc.len = q - q0;
            f->scans.push_back(sc);
            p = q;
            continue;
        }
        p += L;
    }
    if (!have_frame || f->scans.empty()) return JPG_BAD;
    for (int c = 0; c < f->ncomp; c++) if (!latched[c]) return JPG_BAD;          // a component that no scan mentions
    if (f->progressive)
        for (int c = 0; c < f->ncomp; c++)
            for (int k = 0; k < 10; k++) if (cbits[c][k] != 0) return JPG_UNSUPPORTED;   // libjpeg would smooth this image
    return JPG_OK;
}

struct JpegBits {
    const uint8_t* d; size_t n, p = 0;
    uint64_t acc = 0; int cnt = 0;
    bool hit_marker = false;
    long long real_bits = 0, used_bits = 0;      // bits delivered from the file / consumed since the last restart
    I2S_HD JpegBits(const uint8_t* d_, size_t n_) : d(d_), n(n_) {}
    I2S_HD inline void fill()
    {
        // fast path: four data bytes without 0xFF (no stuffing, no marker) enter the accumulator at once
        while (cnt <= 32 && !hit_marker && p + 4 <= n) {
            const uint32_t w = ((uint32_t)d[p] << 24) | ((uint32_t)d[p + 1] << 16) | ((uint32_t)d[p + 2] << 8) | d[p + 3];
            const uint32_t inv = ~w;                                 // a byte of w is 0xFF iff the byte of inv is 0
            if (((inv - 0x01010101u) & ~inv & 0x80808080u) != 0) break;
            acc = (acc << 32) | w;
            cnt += 32; p += 4; real_bits += 32;
        }
        while (cnt <= 56) {
            unsigned b = 0;
            if (!hit_marker && p < n) {
                b = d[p];
                if (b == 0xFF) {
                    const unsigned nx = p + 1 < n ? d[p + 1] : 0xD9;
                    if (nx == 0) { p += 2; real_bits += 8; }
                    else { hit_marker = true; b = 0; }                // feed zeros at a marker (as libjpeg does)
                } else { p++; real_bits += 8; }
            }
            acc = (acc << 8) | b;
            cnt += 8;
        }
    }
    I2S_HD inline unsigned peek(int k) { if (cnt < k) fill(); return (unsigned)(acc >> (cnt - k)) & ((1u << k) - 1u); }
    I2S_HD inline void skip(int k) { cnt -= k; used_bits += k; }
    I2S_HD inline bool overrun() const { return used_bits > real_bits; }     // consumed padding zeros: the segment ended early
    I2S_HD inline unsigned get(int k) { if (k == 0) return 0; const unsigned v = peek(k); skip(k); return v; }
    // byte-align and step over the RSTn marker
    I2S_HD inline bool restart()
    {
        if (overrun()) return false;
        acc = 0; cnt = 0; real_bits = 0; used_bits = 0;
        if (!hit_marker) {                                           // skip what is left of the interval, up to the marker
            while (p + 1 < n && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) p++;
        }
        while (p + 2 < n && d[p] == 0xFF && d[p + 1] == 0xFF) p++;   // fill bytes in front of the marker (B.1.1.2): any number of FF
        if (p + 1 >= n || d[p] != 0xFF || d[p + 1] < 0xD0 || d[p + 1] > 0xD7) return false;
        p += 2;
        hit_marker = false;
        return true;
    }
};

I2S_HD static inline int jpg_decode_sym(JpegBits& b, const JpegHuff& h)
{
    const unsigned look = b.peek(9);
    const int l = h.look_len[look];
    if (l) { b.skip(l); return h.look_sym[look]; }
    int code = (int)b.get(9), len = 9;
    while (len < 17 && code > h.maxcode[len]) { code = (code << 1) | (int)b.get(1); len++; }
    if (len > 16) return -1;
    return h.syms[(code + h.valoff[len]) & 255];
}

I2S_HD static inline int jpg_extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

// One block of a sequential scan.
I2S_HD static inline int jpg_block_sequential(JpegBits& b, const JpegHuff& hd, const JpegHuff& ha, int& pred, int16_t* blk)
{
    int s = jpg_decode_sym(b, hd);
    if (s < 0 || s > 11) return JPG_BAD;
    pred += s ? jpg_extend((int)b.get(s), s) : 0;
    blk[0] = (int16_t)pred;
    for (int k = 1; k < 64;) {
        const int rs = jpg_decode_sym(b, ha);
        if (rs < 0) return JPG_BAD;
        const int r = rs >> 4;
        s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return JPG_BAD;
            blk[JPG_ZZ_AT(k)] = (int16_t)jpg_extend((int)b.get(s), s);
            k++;
        } else if (r == 15) k += 16;
        else break;
    }
    return JPG_OK;
}

// Progressive AC, first pass of a band (jdphuff.c decode_mcu_AC_first).
I2S_HD static inline int jpg_block_ac_first(JpegBits& b, const JpegHuff& ha, int ss, int se, int al, unsigned& eobrun, int16_t* blk)
{
    if (eobrun > 0) { eobrun--; return JPG_OK; }
    for (int k = ss; k <= se; k++) {
        const int rs = jpg_decode_sym(b, ha);
        if (rs < 0) return JPG_BAD;
        const int r = rs >> 4, s = rs & 15;
        if (s) {
            k += r;
            if (k > 63) return JPG_BAD;
            blk[JPG_ZZ_AT(k)] = (int16_t)(jpg_extend((int)b.get(s), s) * (1 << al));
        } else if (r == 15) k += 15;
        else {
            eobrun = 1u << r;
            if (r) eobrun += b.get(r);
            eobrun--;                                               // this block is the first of the run
            break;
        }
    }
    return JPG_OK;
}

// Progressive AC, refinement pass (jdphuff.c decode_mcu_AC_refine): one more bit for the coefficients that are already
// non-zero, and newly non-zero coefficients (+-1 << al) placed after skipping r zero-valued positions.
I2S_HD static inline int jpg_block_ac_refine(JpegBits& b, const JpegHuff& ha, int ss, int se, int al, unsigned& eobrun, int16_t* blk)
{
    const int p1 = 1 << al, m1 = -(1 << al);
    int k = ss;
    if (eobrun == 0) {
        for (; k <= se; k++) {
            const int rs = jpg_decode_sym(b, ha);
            if (rs < 0) return JPG_BAD;
            int r = rs >> 4, s = rs & 15;
            if (s) {
                if (s != 1) return JPG_BAD;
                s = b.get(1) ? p1 : m1;
            } else if (r != 15) {
                eobrun = 1u << r;
                if (r) eobrun += b.get(r);
                break;                                              // the rest of this block is handled as part of the run
            }
            // advance over already-non-zero coefficients (each takes a correction bit) and r zero ones
            do {
                int16_t* c = blk + JPG_ZZ_AT(k);
                if (*c != 0) {
                    if (b.get(1) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
                } else if (--r < 0) break;
                k++;
            } while (k <= se);
            if (s) {
                if (k > 63) return JPG_BAD;
                blk[JPG_ZZ_AT(k)] = (int16_t)s;
            }
        }
    }
    if (eobrun > 0) {
        for (; k <= se; k++) {
            int16_t* c = blk + JPG_ZZ_AT(k);
            if (*c != 0 && b.get(1) && (*c & p1) == 0) *c = (int16_t)(*c + (*c >= 0 ? p1 : m1));
        }
        eobrun--;
    }
    return JPG_OK;
}

// What the entropy decoder needs to know about a frame and about one scan (plain data: the device gets it in this form).
struct JpegFrameView {
    int progressive, ncomp, mcux, mcuy;
    JpegComp c[3];
};
struct JpegScanView {
    int ns, ci[3], td[3], ta[3], ss, se, ah, al, dri;
    const JpegHuff* dc[4];
    const JpegHuff* ac[4];
    const uint8_t* data;
    size_t len;
};

// One scan.  coef[c]: bh * bw blocks of 64 int16 in natural order, zero-initialised before the first scan.
I2S_HD static inline int jpg_decode_scan_view(const JpegFrameView& f, const JpegScanView& sc, int16_t* const coef[3])
{
    JpegBits b(sc.data, sc.len);
    int pred[3] = {0, 0, 0};
    unsigned eobrun = 0;
    long long cnt = 0;
    // a scan of one component is not interleaved: its MCU is one block and it covers only the blocks that hold image samples
    const bool single = sc.ns == 1;
    const JpegComp& c0 = f.c[sc.ci[0]];
    const int nx = single ? (c0.dw + 7) / 8 : f.mcux, ny = single ? (c0.dh + 7) / 8 : f.mcuy;
    for (int my = 0; my < ny; my++)
        for (int mx = 0; mx < nx; mx++) {
            if (sc.dri && cnt && cnt % sc.dri == 0) {
                if (!b.restart()) return JPG_BAD;
                pred[0] = pred[1] = pred[2] = 0;
                eobrun = 0;
            }
            cnt++;
            for (int k = 0; k < sc.ns; k++) {
                const JpegComp& jc = f.c[sc.ci[k]];
                const int nbx = single ? 1 : jc.h, nby = single ? 1 : jc.v;
                for (int by = 0; by < nby; by++)
                    for (int bx = 0; bx < nbx; bx++) {
                        const int row = single ? my : my * jc.v + by, col = single ? mx : mx * jc.h + bx;
                        int16_t* blk = coef[sc.ci[k]] + ((size_t)row * jc.bw + (size_t)col) * 64;
                        int rc = JPG_OK;
                        if (!f.progressive) rc = jpg_block_sequential(b, *sc.dc[sc.td[k]], *sc.ac[sc.ta[k]], pred[k], blk);
                        else if (sc.ss == 0) {
                            if (sc.ah == 0) {                    // DC first pass: the difference, scaled
                                const int s = jpg_decode_sym(b, *sc.dc[sc.td[k]]);
                                if (s < 0 || s > 11) return JPG_BAD;
                                pred[k] += s ? jpg_extend((int)b.get(s), s) : 0;
                                blk[0] = (int16_t)(pred[k] * (1 << sc.al));
                            } else if (b.get(1)) blk[0] = (int16_t)(blk[0] | (1 << sc.al));      // DC refinement: one bit
                        } else if (sc.ah == 0) rc = jpg_block_ac_first(b, *sc.ac[sc.ta[k]], sc.ss, sc.se, sc.al, eobrun, blk);
                        else rc = jpg_block_ac_refine(b, *sc.ac[sc.ta[k]], sc.ss, sc.se, sc.al, eobrun, blk);
                        if (rc) return rc;
                    }
            }
        }
    return b.overrun() ? JPG_BAD : JPG_OK;
}

static inline JpegFrameView jpg_frame_view(const JpegFile& f)
{
    JpegFrameView v;
    v.progressive = f.progressive ? 1 : 0; v.ncomp = f.ncomp; v.mcux = f.mcux; v.mcuy = f.mcuy;
    for (int c = 0; c < 3; c++) v.c[c] = f.c[c];
    return v;
}

// Host half of the parallel entropy decoder (k_jpeg_entropy.h).  A `clean` scan is cut at its RSTn markers into restart
// intervals; jpg_interval gives interval i's raw bytes and its size once the stuffing is removed, jpg_destuff removes it
// (FF00 -> FF; one memchr per 0xFF byte, a few GB/s).
static inline void jpg_interval(const JpegScan& sc, size_t i, size_t* raw0, size_t* raw_len, size_t* nbytes)
{
    const size_t a = i == 0 ? 0 : sc.rst[i - 1].pos + 2, b = i < sc.rst.size() ? sc.rst[i].pos : sc.len;
    const size_t st = (i < sc.rst.size() ? sc.rst[i].nstuff : sc.nstuff) - (i == 0 ? 0 : sc.rst[i - 1].nstuff);
    *raw0 = a; *raw_len = b - a; *nbytes = b - a - st;
}
static inline size_t jpg_destuff(const uint8_t* d, size_t n, uint8_t* out)
{
    size_t p = 0, o = 0;
    while (p < n) {
        const uint8_t* ff = static_cast<const uint8_t*>(memchr(d + p, 0xFF, n - p));
        const size_t q = ff ? (size_t)(ff - d) + 1 : n;           // the FF itself is data
        memcpy(out + o, d + p, q - p);
        o += q - p;
        p = ff ? q + 1 : n;                                         // and the 00 after it is not
    }
    return o;
}

// Host path: the scans of a parsed file from scan `first` on, in file order (first = 0: the whole file, onto zeroed coefficients).
static int jpg_decode_scans_from(const JpegFile& f, int16_t* const coef[3], size_t first)
{
    const JpegFrameView fv = jpg_frame_view(f);
    for (size_t si = first; si < f.scans.size(); si++) {
        const JpegScan& sc = f.scans[si];
        JpegScanView v;
        v.ns = sc.ns; v.ss = sc.ss; v.se = sc.se; v.ah = sc.ah; v.al = sc.al; v.dri = sc.dri;
        for (int k = 0; k < 3; k++) { v.ci[k] = sc.ci[k]; v.td[k] = sc.td[k]; v.ta[k] = sc.ta[k]; }
        for (int t = 0; t < 4; t++) { v.dc[t] = &sc.dc[t]; v.ac[t] = &sc.ac[t]; }
        v.data = sc.data; v.len = sc.len;
        const int rc = jpg_decode_scan_view(fv, v, coef);
        if (rc) return rc;
    }
    return JPG_OK;
}
static int jpg_decode_scan(const JpegFile& f, int16_t* const coef[3]) { return jpg_decode_scans_from(f, coef, 0); }

}  // namespace i2s
