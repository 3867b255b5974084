// Huffman decoding of sequential JPEG scans on the device, parallel INSIDE a scan (the reference reaches the
// pixels through PIL.Image.open(path).convert("RGB"), img2sgf.py:651).
//
// A Huffman stream is serial: where a symbol starts depends on every symbol before it.  But JPEG's codes resynchronise: a
// decoder started at a wrong position almost always falls into step with the true parse within a few dozen symbols.  That is
// used here the way Klein & Wiseman ("Parallel Huffman decoding with applications to JPEG files", 2003) and Weissenberger &
// Schmidt ("Accelerating JPEG decompression on GPUs", 2021) describe it, restated as a fixed-point iteration:
//
//   * the host removes the byte stuffing (FF00 -> FF) and cuts the scan at its RSTn markers into SEGMENTS (restart
//     intervals; a scan without DRI is one segment) -- one memchr-speed copy on the host threads (jpg_destuff, jpeg_host.h);
//   * a segment is cut into SUBSEQUENCES of 1024 bits, one per thread.  The decoder state between two code words is
//     (bit position p, zigzag index z inside the block, block index u inside the MCU).  E[i], the state at the first code-word
//     boundary at or after the start of subsequence i, is exact for i = 0 (0, 0, 0) and GUESSED (start bit, 0, 0) elsewhere;
//   * k_je_sync, round r: thread i decodes subsequence i from E[i] up to the first boundary at or beyond its end -- X_i(E[i]) --
//     and stores that as E[i+1] if it differs, which schedules thread i+1 for round r+1.  When a round changes nothing,
//     E[i+1] = X_i(E[i]) holds for every i and E[0] is exact, so every E[i] is exact by induction: the result does NOT
//     depend on the streams resynchronising, only the number of rounds does (7-9 on rendered diagrams, 15-17 on the reference's scans: a
//     run of identical blocks can hold a false parse in step with the true one and costs one round per subsequence of the run);
//   * each run also counts the blocks it completed and sums the DC differences per component; k_je_scan turns them into the
//     block ordinal and the DC predictors at the start of every subsequence (one wave per segment, running sums);
//   * k_je_write decodes every subsequence once more from its exact state and stores the coefficients where they belong.
//
// The verdict matches jpg_decode_scan_view (jpeg_host.h): a segment is good iff the true parse completes at least the blocks
// its restart interval holds before it runs out of bits (what follows them is ignored, as the serial decoder never looks at
// it).  Progressive files (round 3): the scans in front of the first refinement pass -- DC first passes and AC first passes, the
// same resynchronising code words with an end-of-band run on top -- are decoded here as well; the refinement passes, whose parse
// depends on which coefficients of a block are already non-zero, stay with the serial decoder on the host threads, which takes
// over the coefficient arrays where the device left them (api_jpeg.h).
#pragma once
#include "i2s_types.h"
#include "jpeg_host.h"

namespace i2s {

constexpr int JE_SUB_BYTES = 128, JE_SUB_BITS = 1024;     // one subsequence
constexpr int JE_BLOCK = 256;                             // subsequences (= threads) per workgroup, all of ONE scan
constexpr int JE_SLOT = 35;                               // LDS dwords per thread: 32 + what a code word starting at bit 1023 may reach, odd stride
constexpr int JE_TAB_DW = (int)(sizeof(JpegHuff) / 4);
static_assert(sizeof(JpegHuff) % 4 == 0, "tables are staged dword by dword");
constexpr unsigned long long JE_DEAD = ~0ull;             // the parse ran out of bits: nothing follows

struct JeSeg {
    uint32_t off, nbytes;        // destuffed bytes of the restart interval inside the pass's blob (off is a multiple of 4)
    uint32_t sub0;               // its first subsequence (global numbering of the pass)
    uint32_t blk0, nblk;         // first block ordinal inside the scan; the blocks the interval must hold
    int32_t file;                // image of the pass (status)
};

struct JeScan {
    int16_t* coef[3];            // per component SLOT k of the scan: blocks of 64 coefficients, natural order
    int bw[3], hk[3], vk[3];     // blocks per row of the component's array; blocks per MCU (1 x 1 in a single-component scan)
    int tab_dc[3], tab_ac[3];    // the slot's tables in the pass's pool
    int ns, bpm, nx;             // components in the scan, blocks per MCU, MCUs per row
    int seg0, nseg;              // its segments in segs[]
    uint32_t sub0, nsub;         // its subsequences (sub0 is a multiple of JE_BLOCK)
    // 0: a sequential scan.  Progressive files, first passes only (Ah = 0; a refinement pass does not resynchronise, see below):
    // 1: DC scan -- every code word is a DC difference and completes its block; 2: AC scan of ONE component, band ss .. se -- a
    // block starts at zigzag index ss, EOBn ends the block AND the next 2^n + bits - 1 ones.  Values are scaled by 2^al.
    int kind, ss, se, al;
};

// blocks completed, sum of the DC differences per component slot, blocks completed when the first impossible code word was met (-1: none)
struct JeAcc { int cnt, dc[3], bad_at; };

__device__ __forceinline__ unsigned long long je_pack(uint32_t p, int z, int u) { return ((unsigned long long)p << 16) | ((unsigned)z << 8) | (unsigned)u; }

struct JeShared {
    uint32_t tab[6][JE_TAB_DW];             // DC tables of slots 0..2, AC tables of slots 0..2
    uint32_t slot[JE_BLOCK * JE_SLOT];      // the threads' bits, big-endian dwords
    uint8_t zz[64];
    uint8_t uk[8], ubx[8], uby[8];          // block u of the MCU: component slot, block column / row inside the MCU
    JeScan sc;
    int any;
};

// the segment of scan sc that holds global subsequence g
__device__ __forceinline__ int je_find_seg(const JeSeg* __restrict__ segs, const JeScan& sc, uint32_t g)
{
    int lo = sc.seg0, hi = sc.seg0 + sc.nseg - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (segs[mid].sub0 <= g) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__device__ __forceinline__ void je_stage(JeShared& sh, const JeScan& sc, const JpegHuff* __restrict__ tabs, const uint32_t* __restrict__ blob,
                                         uint32_t my_off_dw, bool active)
{
    const int tid = threadIdx.x;
    if (tid == 64) sh.sc = sc;
    for (int t = 0; t < sc.ns; t++) {
        const uint32_t* d = reinterpret_cast<const uint32_t*>(tabs + sc.tab_dc[t]);
        const uint32_t* a = reinterpret_cast<const uint32_t*>(tabs + sc.tab_ac[t]);
        for (int i = tid; i < JE_TAB_DW; i += JE_BLOCK) { sh.tab[t][i] = d[i]; sh.tab[3 + t][i] = a[i]; }
    }
    if (tid < 64) sh.zz[tid] = JPG_ZZ_AT(tid);
    if (tid == 0) {
        int u = 0;
        for (int k = 0; k < sc.ns; k++)
            for (int by = 0; by < sc.vk[k]; by++)
                for (int bx = 0; bx < sc.hk[k]; bx++, u++) { sh.uk[u & 7] = (uint8_t)k; sh.ubx[u & 7] = (uint8_t)bx; sh.uby[u & 7] = (uint8_t)by; }
    }
    // every wave copies the 33 dwords of each of its lanes' subsequences with one coalesced load (lane = dword)
    const int lane = tid & 63, w0 = tid & ~63;
    for (int t = 0; t < 64; t++) {
        const uint32_t o = __shfl(my_off_dw, t);
        const int act = __shfl(active ? 1 : 0, t);
        if (act && lane < JE_SLOT - 2) sh.slot[(w0 + t) * JE_SLOT + lane] = __builtin_bswap32(blob[o + lane]);
    }
    __syncthreads();
}

// One code word with its value bits: 32 bits at slot bit `pos` hold both (a code is at most 16 bits, a value at most 15).
__device__ __forceinline__ uint32_t je_peek32(const uint32_t* slot, uint32_t pos)
{
    const uint32_t i = pos >> 5, sh = pos & 31;
    const unsigned long long w = ((unsigned long long)slot[i] << 32) | slot[i + 1];
    return (uint32_t)(w >> (32 - sh));
}

__device__ __forceinline__ int je_symbol(const JpegHuff& h, uint32_t v, int& len)
{
    const uint32_t look = v >> 23;
    const int l = h.look_len[look], sm = h.look_sym[look];     // both reads leave together
    if (l) { len = l; return sm; }
    int code = (int)look, n = 9;
    while (n < 17 && code > h.maxcode[n]) { code = (code << 1) | (int)((v >> (31 - n)) & 1u); n++; }
    len = n;
    if (n > 16) return -1;
    return h.syms[(code + h.valoff[n]) & 255];
}

// The decoder proper.  Runs code words from state (p, z, u) while p < end (segment bit positions; slot bit 0 is segment bit
// bit0, L = bits in the segment).  WRITE: coefficients go to the scan's arrays (q = ordinal of the current block inside the
// scan, quota = first ordinal beyond the segment, pred = the DC predictors); otherwise blocks and DC differences are only counted.
template <bool WRITE>
__device__ __forceinline__ unsigned long long je_decode(const JeShared& sh, const uint32_t* slot, uint32_t bit0, uint32_t end,
                                                        uint32_t L, unsigned long long state, JeAcc& acc, uint32_t q, uint32_t quota,
                                                        int* redo = nullptr)
{
    if (state == JE_DEAD) return JE_DEAD;
    const JeScan& sc = sh.sc;
    const int kind = sc.kind, z0 = kind == 2 ? sc.ss : 0, zend = kind == 2 ? sc.se : 63, al = sc.al;
    uint32_t p = (uint32_t)(state >> 16);
    int z = (int)(state >> 8) & 255, u = (int)state & 255;
    int k = sh.uk[u];
    int16_t* blk = nullptr;
    uint32_t mx = 0, my = 0;
    auto place = [&]() {
        const uint32_t m = q / (uint32_t)sc.bpm;
        my = m / (uint32_t)sc.nx; mx = m - my * (uint32_t)sc.nx;
        blk = sc.coef[k] + ((size_t)(my * sc.vk[k] + sh.uby[u]) * sc.bw[k] + (size_t)(mx * sc.hk[k] + sh.ubx[u])) * 64;
    };
    if (WRITE) place();
    while (p < end) {
        if (WRITE && q >= quota) break;                         // the interval's blocks are complete: what follows is not looked at
        const uint32_t v = je_peek32(slot, p - bit0);
        const bool dc = kind != 2 && z == 0;
        const JpegHuff& h = *reinterpret_cast<const JpegHuff*>(sh.tab[dc ? k : 3 + k]);
        // A guessed parse meets code words no encoder writes (no such code, a DC category beyond 11, a run past the end of the
        // block).  It must carry on all the same -- by any fixed rule -- to fall into step with the true parse; the true parse
        // meeting one means a corrupt file: the first is noted and k_je_scan judges whether it lies inside the interval's blocks.
        int len;
        int sym = je_symbol(h, v, len);
        bool bad = false;
        if (sym < 0) { sym = 0; len = 16; bad = true; }
        bool done = false, eob = false;
        int s, zi = z;
        if (dc) {
            bad |= sym > 11;
            s = sym & 15;
            z = 1;
            done = kind == 1;                                   // a progressive DC scan: the block is complete
        } else {
            const int r = sym >> 4;
            s = sym & 15;
            if (s) {
                zi = z + r;
                z = zi + 1;
                if (zi > 63) { bad = true; zi = 64; z = 64; }
            } else if (r == 15) z += 16;
            else {
                z = 64;
                if (kind == 2) { eob = true; s = r; }           // EOBn: r more bits say how long the run of finished blocks is
            }
            done = z > zend;
        }
        if (bad && acc.bad_at < 0) acc.bad_at = acc.cnt;
        if (p + len + s > L) return JE_DEAD;                    // the code word is cut off by the end of the data
        p += len + s;
        int val = 0;
        uint32_t run = 1;                                       // blocks this code word completes
        if (s) {
            const int bits = (int)((v << len) >> (32 - s));
            val = bits < (1 << (s - 1)) ? bits - (1 << s) + 1 : bits;
            if (eob) run = (1u << s) + (uint32_t)bits;
        }
        if (dc) {
            acc.dc[0] += k == 0 ? val : 0; acc.dc[1] += k == 1 ? val : 0; acc.dc[2] += k == 2 ? val : 0;
            if (WRITE) blk[0] = (int16_t)((k == 0 ? acc.dc[0] : (k == 1 ? acc.dc[1] : acc.dc[2])) * (1 << al));
        } else if (WRITE && !eob && s && zi < 64) {
            blk[sh.zz[zi]] = (int16_t)(val * (1 << al));
            // A run that carries past the scan's band (libjpeg stores the coefficient all the same, jdphuff.c decode_mcu_AC_first) lands
            // in a band another scan of this pass may be writing: whose value stays would depend on the scheduling.  Such a
            // (non-conforming) file goes to the serial decoder, which applies the scans in file order.
            if (zi > zend && redo) atomicCAS(redo, (int)JPG_OK, (int)JPG_REDO);
        }
        if (done) {
            z = z0;
            acc.cnt += (int)run;
            if (++u == sc.bpm) u = 0;
            k = sh.uk[u];
            if (WRITE) {
                q += run;
                if (run == 1) {
                    if (u == 0 && ++mx == (uint32_t)sc.nx) { mx = 0; my++; }
                    blk = sc.coef[k] + ((size_t)(my * sc.vk[k] + sh.uby[u]) * sc.bw[k] + (size_t)(mx * sc.hk[k] + sh.ubx[u])) * 64;
                } else place();                                 // (only in single-component scans: u stays 0)
            }
        }
    }
    return je_pack(p, z, u);
}

// grid (blocks of the pass), block JE_BLOCK.  Round `round` of the iteration: stamp[g] == round marks the subsequences whose
// entry state changed in the round before (round 0 runs them all from the guess).  flags[round] is set if anything changed.
__global__ __launch_bounds__(JE_BLOCK) void k_je_sync(const JeScan* __restrict__ scans, const JeSeg* __restrict__ segs, const int* __restrict__ blk_scan,
                                                       const JpegHuff* __restrict__ tabs, const uint32_t* __restrict__ blob, unsigned long long* __restrict__ E,
                                                       uint32_t* __restrict__ stamp, JeAcc* __restrict__ accs, uint32_t* __restrict__ flags, uint32_t round)
{
    __shared__ JeShared sh;
    const JeScan& sc = scans[blk_scan[blockIdx.x]];
    const uint32_t g = blockIdx.x * JE_BLOCK + threadIdx.x;
    bool active = g - sc.sub0 < sc.nsub;
    // after round 0 the stamp alone says who runs (the host zeroes the stamps, and nobody ever stamps the first subsequence
    // of a segment, whose entry state is exact from the start): no segment look-up for the others
    if (round > 0 && active && stamp[g] != round) active = false;
    uint32_t j = 0, off_dw = 0, L = 0;
    bool first = true, last = true;
    if (active) {
        const JeSeg& sg = segs[je_find_seg(segs, sc, g)];
        j = g - sg.sub0;
        first = j == 0;
        L = sg.nbytes * 8;
        last = (j + 1) * JE_SUB_BITS >= L;
        off_dw = sg.off / 4 + j * (JE_SUB_BYTES / 4);
    }
    if (threadIdx.x == 0) sh.any = 0;
    __syncthreads();
    if (active) sh.any = 1;
    __syncthreads();
    if (!sh.any) return;                                        // nothing of this workgroup changed in the round before
    je_stage(sh, sc, tabs, blob, off_dw, active);
    if (!active) return;
    const int z0 = sc.kind == 2 ? sc.ss : 0;                    // zigzag index at the start of a block
    const unsigned long long entry = first ? je_pack(0, z0, 0) : (round == 0 ? je_pack(j * JE_SUB_BITS, z0, 0) : E[g]);
    JeAcc acc = {0, {0, 0, 0}, -1};
    const uint32_t end = last ? L : (j + 1) * JE_SUB_BITS;
    const unsigned long long exit = je_decode<false>(sh, sh.slot + threadIdx.x * JE_SLOT, j * JE_SUB_BITS, end, L, entry, acc, 0, 0);
    accs[g] = acc;
    if (!last && (round == 0 || E[g + 1] != exit)) {
        E[g + 1] = exit;
        stamp[g + 1] = round + 1;
        flags[round] = 1;
    }
}

// grid (blocks of the pass), block JE_BLOCK: pend[file] = 1 for the files that still have a subsequence scheduled for round `round`
// (run when the iteration stops at its limit: only those files go to the serial decoder, the others have reached their fixed point)
__global__ __launch_bounds__(JE_BLOCK) void k_je_pending(const JeScan* __restrict__ scans, const JeSeg* __restrict__ segs, const int* __restrict__ blk_scan,
                                                          const uint32_t* __restrict__ stamp, uint32_t round, int* __restrict__ pend)
{
    const JeScan& sc = scans[blk_scan[blockIdx.x]];
    const uint32_t g = blockIdx.x * JE_BLOCK + threadIdx.x;
    if (g - sc.sub0 >= sc.nsub || stamp[g] != round) return;
    pend[segs[je_find_seg(segs, sc, g)].file] = 1;
}

// grid (segments of the pass / 4), block 256: one wave per segment.  base[g] = blocks completed and DC sums before subsequence
// g (running sums of accs over the segment); a segment whose parse holds fewer blocks than its interval marks the file bad.
// skip[file] != 0: the iteration stopped at its round limit with work of this file still scheduled (k_je_pending) -- its states are not
// at their fixed point, the file goes to the serial decoder, and neither a verdict nor a coefficient of it is written here.
__global__ __launch_bounds__(256) void k_je_scan(const JeSeg* __restrict__ segs, int nseg, const JeAcc* __restrict__ accs, JeAcc* __restrict__ base,
                                                 int* __restrict__ status, const int* __restrict__ skip)
{
    const int s = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (s >= nseg) return;
    const JeSeg sg = segs[s];
    if (skip[sg.file]) return;                                   // wave-uniform: one segment per wavefront
    const uint32_t n = sg.nbytes == 0 ? 1u : (sg.nbytes + JE_SUB_BYTES - 1) / JE_SUB_BYTES;
    JeAcc carry = {0, {0, 0, 0}, -1};
    bool bad = false;
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        JeAcc a = {0, {0, 0, 0}, -1};
        if (i < n) a = accs[sg.sub0 + i];
        JeAcc inc = a;
        for (int d = 1; d < 64; d <<= 1) {
            const int c = __shfl_up(inc.cnt, d), d0 = __shfl_up(inc.dc[0], d), d1 = __shfl_up(inc.dc[1], d), d2 = __shfl_up(inc.dc[2], d);
            if (lane >= d) { inc.cnt += c; inc.dc[0] += d0; inc.dc[1] += d1; inc.dc[2] += d2; }
        }
        if (i < n) {
            JeAcc b;
            b.cnt = carry.cnt + inc.cnt - a.cnt;
            for (int c = 0; c < 3; c++) b.dc[c] = carry.dc[c] + inc.dc[c] - a.dc[c];
            b.bad_at = -1;
            base[sg.sub0 + i] = b;
            bad |= a.bad_at >= 0 && (uint32_t)(b.cnt + a.bad_at) < sg.nblk;     // an impossible code word inside a block that counts
        }
        carry.cnt += __shfl(inc.cnt, 63);
        for (int c = 0; c < 3; c++) carry.dc[c] += __shfl(inc.dc[c], 63);
    }
    if (__any(bad ? 1 : 0) || (uint32_t)carry.cnt < sg.nblk) {
        if (lane == 0) status[sg.file] = JPG_BAD;
    }
}

// grid / block as k_je_sync: every subsequence once more, from its exact entry state, coefficients stored.
__global__ __launch_bounds__(JE_BLOCK) void k_je_write(const JeScan* __restrict__ scans, const JeSeg* __restrict__ segs, const int* __restrict__ blk_scan,
                                                        const JpegHuff* __restrict__ tabs, const uint32_t* __restrict__ blob,
                                                        const unsigned long long* __restrict__ E, const JeAcc* __restrict__ base, int* __restrict__ status,
                                                        const int* __restrict__ skip)
{
    __shared__ JeShared sh;
    const JeScan& sc = scans[blk_scan[blockIdx.x]];
    const uint32_t g = blockIdx.x * JE_BLOCK + threadIdx.x;
    const bool active = g - sc.sub0 < sc.nsub;
    uint32_t j = 0, off_dw = 0, L = 0, blk0 = 0, nblk = 0;
    int file = 0;
    if (active) {
        const JeSeg& sg = segs[je_find_seg(segs, sc, g)];
        j = g - sg.sub0;
        L = sg.nbytes * 8;
        off_dw = sg.off / 4 + j * (JE_SUB_BYTES / 4);
        blk0 = sg.blk0; nblk = sg.nblk; file = sg.file;
    }
    je_stage(sh, sc, tabs, blob, off_dw, active);
    if (!active || skip[file]) return;
    const unsigned long long entry = j == 0 ? je_pack(0, sc.kind == 2 ? sc.ss : 0, 0) : E[g];
    JeAcc acc = base[g];
    if ((uint32_t)acc.cnt >= nblk) return;
    const bool last = (j + 1) * JE_SUB_BITS >= L;
    const uint32_t end = last ? L : (j + 1) * JE_SUB_BITS;
    je_decode<true>(sh, sh.slot + threadIdx.x * JE_SLOT, j * JE_SUB_BITS, end, L, entry, acc, blk0 + (uint32_t)acc.cnt, blk0 + nblk, status + file);
}

}  // namespace i2s
