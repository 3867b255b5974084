// JPEG input of the C ABI (include/i2s.h: i2s_jpeg_info, i2s_detect_jpeg_batch and their diagnostics): marker
// parsing on the host threads, the entropy stage of a pass three ways (parallel inside a scan on the device, one lane per
// file, host threads), then k_jpeg_idct / k_jpeg_rgb and the ordinary detection path on the decoded, device-resident images.
// Part of i2s_api.hip (included there, after struct i2s_ctx and the detection entry points it calls).
#pragma once
// ---- JPEG input ------------------------------------------------------------------------------------------------

extern "C" int i2s_jpeg_info(const uint8_t* data, size_t len, int* w, int* h, int* channels)
{
    if (!data) return I2S_E_INVALID;
    try {
        JpegFile f;
        const int rc = jpg_parse(data, len, &f);
        if (rc == JPG_BAD) return I2S_E_INVALID;
        if (rc == JPG_UNSUPPORTED) return I2S_E_UNSUPPORTED;
        if (w) *w = f.X;
        if (h) *h = f.Y;
        if (channels) *channels = f.ncomp;
        return I2S_OK;
    } catch (const std::exception&) {
        return I2S_E_INVALID;
    }
}

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

int JpegCoefHost::reserve(i2s_ctx* ctx, size_t count)
{
    if (count <= n) return I2S_OK;
    I2S_HIP(hipStreamSynchronize(ctx->stream));
    if (p) I2S_HIP(hipHostFree(p));
    p = nullptr; n = 0;
    const size_t want = count + count / 4;
    I2S_HIP(hipHostMalloc(&p, want * sizeof(int16_t)));
    n = want;
    return I2S_OK;
}

static inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// fn(i) for i in [0, n) on up to 16 host threads (the calling one included).  fn must not throw.
template <class F>
static void rc_parallel_for(int n, F&& fn)
{
    if (n <= 0) return;
    std::atomic<int> next(0);
    auto work = [&]() { for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    const unsigned hw = std::thread::hardware_concurrency();
    const int nthreads = std::max(1, std::min({n, (int)(hw ? hw : 1), 16}));
    std::vector<std::thread> pool;
    for (int t = 1; t < nthreads; t++) {
        try { pool.emplace_back(work); } catch (const std::exception&) { break; }      // carry on with the threads we have
    }
    work();
    for (auto& t : pool) t.join();
}

// ---- entropy decoding of one pass of JPEG files ---------------------------------------------------------------------------
// The coefficient arrays of the pass (ctx->h_jd[i].coef[c], laid out by the caller inside ctx->d_jpg, ncoef bytes in all) are
// zeroed and filled, three ways:
//   jpeg_parallel                       : sequential files, parallel inside each scan (k_jpeg_entropy.h)
//   jpeg_lanes                          : any file, one lane per file (k_jpeg_huffman)
//   jpeg_host_decode + jpeg_host_upload : any file, one host thread per file (jpeg_host.h), then one upload per component
// mode (i2s_params.jpeg_entropy_device): 0 = all on host threads; 1 = sequential files in parallel on the device, the others
// (progressive, or entropy-coded data with anything but stuffed FF00 and RSTn in it) on host threads; 2 = those on lanes.

static int jpeg_grow(i2s_ctx* ctx, uint8_t** buf, size_t* have, size_t need)
{
    if (need <= *have) return I2S_OK;
    I2S_HIP(hipStreamSynchronize(ctx->stream));
    if (*buf) I2S_HIP(hipFree(*buf));
    *buf = nullptr; *have = 0;
    I2S_HIP(hipMalloc(buf, need));
    *have = need;
    return I2S_OK;
}

static int jpeg_bad(i2s_ctx* ctx, int k)
{
    snprintf(ctx->err, sizeof(ctx->err), "JPEG %d: corrupt or truncated entropy-coded data", k);
    return I2S_E_INVALID;
}

// The serial decoder on the host threads for `list` (no HIP calls: it may run beside the device work); -1 or the input index
// of a corrupt file.  coef must already hold ncoef bytes.
static int jpeg_host_decode(i2s_ctx* ctx, const std::vector<int>& list, const std::vector<JpegFile>& files, const int* order, JpegCoefHost& coef)
{
    const int16_t* d0 = reinterpret_cast<const int16_t*>(ctx->d_jpg);
    std::atomic<int> bad(-1);
    rc_parallel_for((int)list.size(), [&](int n) {
        const int i = list[n];
        const JpegFile& f = files[i];
        int16_t* cp[3] = {nullptr, nullptr, nullptr};
        for (int c = 0; c < f.ncomp; c++) {
            cp[c] = coef.data() + (ctx->h_jd[i].coef[c] - d0);
            memset(cp[c], 0, (size_t)f.c[c].bw * f.c[c].bh * 64 * sizeof(int16_t));
        }
        if (jpg_decode_scan(f, cp) != JPG_OK) bad.store(order[i]);
    });
    return bad.load();
}

// The scans of progressive files that the device does not take (from dev_scans[i] on: the refinement passes and whatever follows
// them), on the host threads, continuing on the coefficient arrays the device produced (already copied into `coef`).
static int jpeg_host_finish(i2s_ctx* ctx, const std::vector<int>& list, const std::vector<JpegFile>& files, const int* order,
                            const std::vector<int>& dev_scans, JpegCoefHost& coef)
{
    const int16_t* d0 = reinterpret_cast<const int16_t*>(ctx->d_jpg);
    std::atomic<int> bad(-1);
    rc_parallel_for((int)list.size(), [&](int n) {
        const int i = list[n];
        const JpegFile& f = files[i];
        int16_t* cp[3] = {nullptr, nullptr, nullptr};
        for (int c = 0; c < f.ncomp; c++) cp[c] = coef.data() + (ctx->h_jd[i].coef[c] - d0);
        if (jpg_decode_scans_from(f, cp, (size_t)dev_scans[i]) != JPG_OK) bad.store(order[i]);
    });
    return bad.load();
}

static int jpeg_host_upload(i2s_ctx* ctx, const std::vector<int>& list, const std::vector<JpegFile>& files, JpegCoefHost& coef)
{
    const int16_t* d0 = reinterpret_cast<const int16_t*>(ctx->d_jpg);
    for (int i : list)
        for (int c = 0; c < files[i].ncomp; c++) {
            const size_t off = (size_t)(ctx->h_jd[i].coef[c] - d0);
            I2S_HIP(hipMemcpyAsync(ctx->d_jpg + off * sizeof(int16_t), coef.data() + off, (size_t)files[i].c[c].bw * files[i].c[c].bh * 64 * sizeof(int16_t),
                                   hipMemcpyHostToDevice, ctx->stream));
        }
    return I2S_OK;
}

// Runs the lane decoder for `list`; the verdicts are in ctx->d_jstatus[i].
static int jpeg_lanes(i2s_ctx* ctx, const std::vector<int>& list, const std::vector<JpegFile>& files, const uint8_t* const* jpeg, const size_t* len,
                      const int* order, std::vector<uint8_t>& bytes)
{
    if (list.empty()) return I2S_OK;
    const int nl = (int)list.size();
    std::vector<JpegHuff> tabs;
    std::vector<JpgHuffScan> hscans;
    std::vector<JpgHuffImg> himgs(nl);
    auto tab_index = [&](const JpegHuff& h) -> int {
        if (!h.present) return -1;
        for (size_t t = tabs.size(); t-- > 0;)                  // newest first: consecutive scans mostly share tables
            if (memcmp(&tabs[t], &h, sizeof(JpegHuff)) == 0) return (int)t;
        tabs.push_back(h);
        return (int)tabs.size() - 1;
    };
    size_t blob = 0;
    for (int i : list) blob += align256(len[order[i]]);
    bytes.resize(blob);
    size_t bo = 0;
    for (int n = 0; n < nl; n++) {
        const int i = list[n], k = order[i];
        const JpegFile& f = files[i];
        memcpy(bytes.data() + bo, jpeg[k], len[k]);
        JpgHuffImg& hi = himgs[n];
        hi.f = jpg_frame_view(f);
        hi.scan0 = (int)hscans.size(); hi.nscans = (int)f.scans.size();
        hi.slot = i;
        for (int c = 0; c < 3; c++) hi.coef[c] = const_cast<int16_t*>(ctx->h_jd[i].coef[c]);
        for (const JpegScan& sc : f.scans) {
            JpgHuffScan hs;
            hs.ns = sc.ns; hs.ss = sc.ss; hs.se = sc.se; hs.ah = sc.ah; hs.al = sc.al; hs.dri = sc.dri;
            for (int q = 0; q < 3; q++) { hs.ci[q] = sc.ci[q]; hs.td[q] = sc.td[q]; hs.ta[q] = sc.ta[q]; }
            for (int t = 0; t < 4; t++) { hs.tab_dc[t] = tab_index(sc.dc[t]); hs.tab_ac[t] = tab_index(sc.ac[t]); }
            hs.off = (unsigned)(bo + (size_t)(sc.data - jpeg[k])); hs.len = (unsigned)sc.len;
            hscans.push_back(hs);
        }
        bo += align256(len[k]);
    }
    if (tabs.empty()) tabs.resize(1);
    const size_t o_tab = align256(blob), o_scan = o_tab + align256(tabs.size() * sizeof(JpegHuff));
    const size_t o_img = o_scan + align256(hscans.size() * sizeof(JpgHuffScan)), need = o_img + align256(nl * sizeof(JpgHuffImg));
    int rc = jpeg_grow(ctx, &ctx->d_jh, &ctx->jh_bytes, need);
    if (rc) return rc;
    I2S_HIP(hipMemcpyAsync(ctx->d_jh, bytes.data(), blob, hipMemcpyHostToDevice, ctx->stream));
    I2S_HIP(hipMemcpyAsync(ctx->d_jh + o_tab, tabs.data(), tabs.size() * sizeof(JpegHuff), hipMemcpyHostToDevice, ctx->stream));
    I2S_HIP(hipMemcpyAsync(ctx->d_jh + o_scan, hscans.data(), hscans.size() * sizeof(JpgHuffScan), hipMemcpyHostToDevice, ctx->stream));
    I2S_HIP(hipMemcpyAsync(ctx->d_jh + o_img, himgs.data(), nl * sizeof(JpgHuffImg), hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_jpeg_huffman, dim3(cdiv(nl, 64)), dim3(64), 0, ctx->stream, reinterpret_cast<const JpgHuffImg*>(ctx->d_jh + o_img),
                       reinterpret_cast<const JpgHuffScan*>(ctx->d_jh + o_scan), reinterpret_cast<const JpegHuff*>(ctx->d_jh + o_tab), ctx->d_jh, nl,
                       ctx->d_jstatus);
    I2S_HIP(hipStreamSynchronize(ctx->stream));                 // the host vectors of this function go out of scope
    return I2S_OK;
}

constexpr int JE_MAX_ROUNDS = 1 << 16;     // room in the flag array; the context's je_max_rounds (48) is the working limit
constexpr int JE_MAX_SEGS = 1 << 16;       // restart intervals per scan handled on the device

// Sequential files of `list`, parallel inside each scan.  Files whose entropy-coded data hold more than stuffed bytes and RSTn
// markers are moved to `others`, and so are -- round 4, per FILE instead of per pass -- the files whose iteration has not reached its
// fixed point at the limit (a blank page: one subsequence per round): the others are done.  On return the kernels have run
// (ctx->d_jstatus holds the verdicts of the files still in `list`).
// dev_scans[i] (out): how many scans of file i the device decodes -- all of a sequential file, the scans in front of the first
// refinement pass of a progressive one (the host threads take over from there, jpeg_host_finish).
static int jpeg_parallel(i2s_ctx* ctx, std::vector<int>& list, std::vector<int>& others, const std::vector<JpegFile>& files, const int* order,
                         std::vector<int>& dev_scans)
{
    std::vector<JpegHuff> tabs;
    std::vector<JeScan> scans;
    std::vector<JeSeg> segs;
    std::vector<const uint8_t*> raw;             // per segment: its bytes in the file
    std::vector<size_t> raw_len;
    std::vector<int> blk_scan, kept;
    std::vector<size_t> seg_first;               // per kept file: its first segment (and one past the last file's)
    auto tab_index = [&](const JpegHuff& h) -> int {
        for (size_t t = tabs.size(); t-- > 0;)
            if (memcmp(&tabs[t], &h, sizeof(JpegHuff)) == 0) return (int)t;
        tabs.push_back(h);
        return (int)tabs.size() - 1;
    };
    // layout: the parser noted where the restart markers are and how many stuffed bytes lie between them, so every segment's
    // place in the blob is known before a byte is copied
    uint32_t nsub = 0;
    size_t bo = 0;
    for (int i : list) {
        const JpegFile& f = files[i];
        bool ok = true;
        size_t bytes = 0;
        // a progressive file: its leading first passes (Ah = 0)
        size_t nd = f.scans.size();
        if (f.progressive) {
            // The device decodes these scans CONCURRENTLY, so they must write disjoint coefficients: a repeated DC first pass or
            // overlapping AC bands -- streams libjpeg accepts with a warning and decodes in file order, last writer wins -- end the
            // device's share at the first scan that touches a coefficient an earlier one owns.  (A run that carries past
            // its band is caught in the write pass itself: JPG_REDO.)
            unsigned long long owned[3] = {0ull, 0ull, 0ull};
            nd = 0;
            while (nd < f.scans.size() && f.scans[nd].ah == 0) {
                const JpegScan& sc = f.scans[nd];
                const unsigned long long band = (sc.se >= 63 ? ~0ull : ((1ull << (sc.se + 1)) - 1ull)) & ~((1ull << sc.ss) - 1ull);
                bool clash = false;
                for (int k = 0; k < sc.ns; k++) clash |= (owned[sc.ci[k]] & band) != 0ull;
                if (clash) break;
                for (int k = 0; k < sc.ns; k++) owned[sc.ci[k]] |= band;
                nd++;
            }
            ok = nd > 0;
        }
        dev_scans[i] = (int)nd;
        for (size_t si = 0; si < nd; si++) {
            const JpegScan& sc = f.scans[si];
            if (!ok) break;
            const bool single = sc.ns == 1;
            const JpegComp& c0 = f.c[sc.ci[0]];
            const long long mcus = (long long)(single ? (c0.dw + 7) / 8 : f.mcux) * (single ? (c0.dh + 7) / 8 : f.mcuy);
            const long long need = sc.dri ? (mcus + sc.dri - 1) / sc.dri : 1;
            ok = sc.clean && need <= JE_MAX_SEGS;
            if (ok && (long long)sc.rst.size() + 1 < need) return jpeg_bad(ctx, order[i]);        // a restart marker is missing
            bytes += sc.len + 4 * (size_t)need;
            // A sequential scan of next to nothing but empty blocks (a blank page: 5.6 bits per block, DC code + EOB; anything drawn
            // costs tens) never resynchronises a guessed parse -- the iteration would advance one subsequence per round up to its
            // limit and hand the file back then (0.7 ms per round: one lane decodes 170 blocks).  Such a file is a few thousand code
            // words for the serial decoder: it goes there at once.  (Progressive files: their DC passes are legitimately this short.)
            if (ok && !f.progressive && !sc.dri) {
                long long bpm = 1;
                if (!single) { bpm = 0; for (int k = 0; k < sc.ns; k++) bpm += (long long)f.c[sc.ci[k]].h * f.c[sc.ci[k]].v; }
                if ((long long)sc.len * 8 < 8 * mcus * bpm && sc.len > (size_t)ctx->je_max_rounds * JE_SUB_BYTES) ok = false;
            }
        }
        if (!ok || bo + bytes >= (1ull << 31)) { others.push_back(i); continue; }
        kept.push_back(i);
        seg_first.push_back(segs.size());
        for (size_t si = 0; si < nd; si++) {
            const JpegScan& sc = f.scans[si];
            JeScan js;
            memset(&js, 0, sizeof(js));
            const bool single = sc.ns == 1;
            js.ns = sc.ns; js.bpm = 0;
            js.kind = !f.progressive ? 0 : (sc.ss == 0 ? 1 : 2); js.ss = sc.ss; js.se = sc.se; js.al = f.progressive ? sc.al : 0;
            for (int k = 0; k < sc.ns; k++) {
                const JpegComp& jc = f.c[sc.ci[k]];
                js.coef[k] = const_cast<int16_t*>(ctx->h_jd[i].coef[sc.ci[k]]);
                js.bw[k] = jc.bw; js.hk[k] = single ? 1 : jc.h; js.vk[k] = single ? 1 : jc.v;
                js.tab_dc[k] = tab_index(sc.dc[sc.td[k]]); js.tab_ac[k] = tab_index(sc.ac[sc.ta[k]]);
                js.bpm += js.hk[k] * js.vk[k];
            }
            const JpegComp& c0 = f.c[sc.ci[0]];
            js.nx = single ? (c0.dw + 7) / 8 : f.mcux;
            const long long mcus = (long long)js.nx * (single ? (c0.dh + 7) / 8 : f.mcuy);
            const long long need = sc.dri ? (mcus + sc.dri - 1) / sc.dri : 1;
            js.seg0 = (int)segs.size(); js.nseg = (int)need;
            nsub = (nsub + JE_BLOCK - 1) / JE_BLOCK * JE_BLOCK;
            js.sub0 = nsub;
            for (long long n = 0; n < need; n++) {
                size_t r0, rl, nbytes;
                jpg_interval(sc, (size_t)n, &r0, &rl, &nbytes);
                JeSeg sg;
                sg.off = (uint32_t)bo; sg.nbytes = (uint32_t)nbytes;
                sg.sub0 = nsub;
                sg.blk0 = (uint32_t)(n * sc.dri * js.bpm);
                sg.nblk = (uint32_t)((sc.dri ? std::min<long long>(sc.dri, mcus - n * sc.dri) : mcus) * js.bpm);
                sg.file = i;
                segs.push_back(sg);
                raw.push_back(sc.data + r0); raw_len.push_back(rl);
                bo = (bo + nbytes + 3) & ~(size_t)3;
                nsub += nbytes == 0 ? 1u : (uint32_t)((nbytes + JE_SUB_BYTES - 1) / JE_SUB_BYTES);
            }
            js.nsub = nsub - js.sub0;
            for (uint32_t b = js.sub0 / JE_BLOCK; b * JE_BLOCK < nsub; b++) blk_scan.push_back((int)scans.size());
            scans.push_back(js);
        }
    }
    seg_first.push_back(segs.size());
    list.swap(kept);
    if (list.empty()) return I2S_OK;
    const uint32_t nblk = (nsub + JE_BLOCK - 1) / JE_BLOCK, ntot = nblk * JE_BLOCK;
    const size_t blob_bytes = bo + 4 * JE_SLOT;                       // the last subsequence's loads reach past the data
    if (blob_bytes > ctx->jblob_bytes) {
        I2S_HIP(hipStreamSynchronize(ctx->stream));
        if (ctx->h_jblob) I2S_HIP(hipHostFree(ctx->h_jblob));
        ctx->h_jblob = nullptr; ctx->jblob_bytes = 0;
        I2S_HIP(hipHostMalloc(&ctx->h_jblob, blob_bytes + blob_bytes / 4));
        ctx->jblob_bytes = blob_bytes + blob_bytes / 4;
    }
    // the stuffing is removed file by file on the host threads, straight into the pinned blob
    uint8_t* hb = ctx->h_jblob;
    std::atomic<int> short_seg(0);
    rc_parallel_for((int)list.size(), [&](int n) {
        for (size_t sgi = seg_first[n]; sgi < seg_first[n + 1]; sgi++) {
            const size_t o = jpg_destuff(raw[sgi], raw_len[sgi], hb + segs[sgi].off);
            if (o != segs[sgi].nbytes) short_seg.store(1);
            for (size_t z = segs[sgi].off + o; z < ((segs[sgi].off + o + 3) & ~(size_t)3); z++) hb[z] = 0;
        }
    });
    if (short_seg.load()) { snprintf(ctx->err, sizeof(ctx->err), "internal: a restart interval's size does not match the parser's"); return I2S_E_INVALID; }
    memset(hb + bo, 0, 4 * JE_SLOT);
    struct BlobView { uint8_t* p; size_t n; uint8_t* data() const { return p; } size_t size() const { return n; } } blob{hb, blob_bytes};
    const size_t o_tab = align256(blob.size()), o_scan = o_tab + align256(tabs.size() * sizeof(JpegHuff));
    const size_t o_seg = o_scan + align256(scans.size() * sizeof(JeScan)), o_bs = o_seg + align256(segs.size() * sizeof(JeSeg));
    const size_t o_E = o_bs + align256(blk_scan.size() * sizeof(int)), o_stamp = o_E + align256((size_t)ntot * 8);
    const size_t o_acc = o_stamp + align256((size_t)ntot * 4), o_base = o_acc + align256((size_t)ntot * sizeof(JeAcc));
    const size_t o_flag = o_base + align256((size_t)ntot * sizeof(JeAcc)), need = o_flag + align256((JE_MAX_ROUNDS + 4) * sizeof(uint32_t));
    int rc = jpeg_grow(ctx, &ctx->d_je, &ctx->je_bytes, need);
    if (rc) return rc;
    uint8_t* D = ctx->d_je;
    hipStream_t st = ctx->stream;
    I2S_HIP(hipMemcpyAsync(D, blob.data(), blob.size(), hipMemcpyHostToDevice, st));
    I2S_HIP(hipMemcpyAsync(D + o_tab, tabs.data(), tabs.size() * sizeof(JpegHuff), hipMemcpyHostToDevice, st));
    I2S_HIP(hipMemcpyAsync(D + o_scan, scans.data(), scans.size() * sizeof(JeScan), hipMemcpyHostToDevice, st));
    I2S_HIP(hipMemcpyAsync(D + o_seg, segs.data(), segs.size() * sizeof(JeSeg), hipMemcpyHostToDevice, st));
    I2S_HIP(hipMemcpyAsync(D + o_bs, blk_scan.data(), blk_scan.size() * sizeof(int), hipMemcpyHostToDevice, st));
    I2S_HIP(hipMemsetAsync(D + o_flag, 0, ((size_t)ctx->je_max_rounds + 16) * sizeof(uint32_t), st));
    I2S_HIP(hipMemsetAsync(D + o_acc, 0, (size_t)ntot * sizeof(JeAcc), st));
    I2S_HIP(hipMemsetAsync(D + o_stamp, 0, (size_t)ntot * sizeof(uint32_t), st));           // no subsequence is scheduled for a round >= 1 yet
    const JeScan* d_scans = reinterpret_cast<const JeScan*>(D + o_scan);
    const JeSeg* d_segs = reinterpret_cast<const JeSeg*>(D + o_seg);
    const int* d_bs = reinterpret_cast<const int*>(D + o_bs);
    const JpegHuff* d_tabs = reinterpret_cast<const JpegHuff*>(D + o_tab);
    const uint32_t* d_blob = reinterpret_cast<const uint32_t*>(D);
    unsigned long long* d_E = reinterpret_cast<unsigned long long*>(D + o_E);
    uint32_t* d_stamp = reinterpret_cast<uint32_t*>(D + o_stamp);
    JeAcc* d_acc = reinterpret_cast<JeAcc*>(D + o_acc);
    JeAcc* d_base = reinterpret_cast<JeAcc*>(D + o_base);
    uint32_t* d_flag = reinterpret_cast<uint32_t*>(D + o_flag);
    // rounds are launched a few at a time (a round with nothing to do costs a launch of workgroups that leave at once), then
    // the flag of the last one is read back
    uint32_t round = 0;
    int burst = 3;
    std::vector<int> handed;                    // files handed back at the limit
    for (;;) {
        const int left = ctx->je_max_rounds - (int)round;
        if (left <= 0) {
            // the limit: which files still have work scheduled?  They go to the serial decoder; every other file's states are final.
            // The mask has its own half of d_jstatus, zeroed by jpeg_entropy_pass, and STAYS: k_je_scan / k_je_write below skip the
            // files in it, so their verdicts stay 0 and their coefficients stay zeroed for the serial decoder (until round 5 the two
            // kernels ran on those files' unconverged entry states and the host reset / wiped what they had written)
            int* d_pend = ctx->d_jstatus + ctx->max_batch;
            hipLaunchKernelGGL(k_je_pending, dim3(nblk), dim3(JE_BLOCK), 0, st, d_scans, d_segs, d_bs, d_stamp, round, d_pend);
            I2S_HIP(hipGetLastError());
            I2S_HIP(hipMemcpyAsync(ctx->h_jstatus, d_pend, (size_t)ctx->max_batch * sizeof(int), hipMemcpyDeviceToHost, st));
            const double t1 = now_ms();
            I2S_HIP(hipStreamSynchronize(st));
            ctx->jpeg_ms[2] += (float)(now_ms() - t1);
            for (size_t k = 0; k < list.size();) {
                if (ctx->h_jstatus[list[k]]) { handed.push_back(list[k]); others.push_back(list[k]); list.erase(list.begin() + (long)k); }
                else k++;
            }
            break;
        }
        for (int n = 0; n < std::min(burst, left); n++, round++)
            hipLaunchKernelGGL(k_je_sync, dim3(nblk), dim3(JE_BLOCK), 0, st, d_scans, d_segs, d_bs, d_tabs, d_blob, d_E, d_stamp, d_acc, d_flag, round);
        I2S_HIP(hipMemcpyAsync(ctx->h_jflag, d_flag + (round - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        const double t0 = now_ms();
        I2S_HIP(hipStreamSynchronize(st));
        ctx->jpeg_ms[2] += (float)(now_ms() - t0);
        if (*ctx->h_jflag == 0) break;
        burst = round < 16 ? 2 : 8;
    }
    ctx->je_rounds = (int)round;
    ctx->je_handed_back = (int)handed.size();
    const int* d_skip = ctx->d_jstatus + ctx->max_batch;      // files handed back at the limit: no verdict, no coefficient is written for them
    hipLaunchKernelGGL(k_je_scan, dim3(cdiv((int)segs.size(), 4)), dim3(256), 0, st, d_segs, (int)segs.size(), d_acc, d_base, ctx->d_jstatus, d_skip);
    hipLaunchKernelGGL(k_je_write, dim3(nblk), dim3(JE_BLOCK), 0, st, d_scans, d_segs, d_bs, d_tabs, d_blob, d_E, d_base, ctx->d_jstatus, d_skip);
    I2S_HIP(hipGetLastError());
    return I2S_OK;
}

static int jpeg_entropy_pass(i2s_ctx* ctx, int nb, const std::vector<JpegFile>& files, const uint8_t* const* jpeg, const size_t* len, const int* order,
                             int mode, size_t ncoef, JpegCoefHost& coef)
{
    std::vector<int> par, host, lanes, late;
    ctx->je_rounds = 0;
    ctx->je_handed_back = 0;
    const double t_in = now_ms();
    const float w_in = ctx->jpeg_ms[2];
    struct Span { i2s_ctx* c; double t; float w; ~Span() { c->jpeg_ms[1] += (float)(now_ms() - t) - (c->jpeg_ms[2] - w); } } span{ctx, t_in, w_in};
    I2S_HIP(hipMemsetAsync(ctx->d_jpg, 0, ncoef, ctx->stream));                      // coefficients start at zero
    I2S_HIP(hipMemsetAsync(ctx->d_jstatus, 0, (size_t)2 * ctx->max_batch * sizeof(int), ctx->stream));     // verdicts and the hand-back mask
    // mode 1: sequential files and the first passes of progressive files on the device (parallel inside every scan), the
    // refinement passes of the latter afterwards on the host threads; mode 2: progressive files one lane per file on the device
    for (int i = 0; i < nb; i++) (mode == 0 ? host : (files[i].progressive && mode == 2 ? lanes : par)).push_back(i);
    // the host threads start on their files at once and run beside the device's
    int host_bad = -1;
    std::thread host_job;
    struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join{host_job};
    // the host copy of coefficient arrays (15 x the file bytes): pinned -- the arrays of progressive files cross the bus twice
    int coef_rc = I2S_OK;
    auto need_coef = [&]() { if (coef_rc == I2S_OK) coef_rc = coef.reserve(ctx, ncoef / sizeof(int16_t)); };
    if (!host.empty()) need_coef();
    if (coef_rc) return coef_rc;
    if (!host.empty()) host_job = std::thread([&]() { host_bad = jpeg_host_decode(ctx, host, files, order, coef); });
    std::vector<int>& rest = mode == 2 ? lanes : late;          // files the parallel decoder hands back
    std::vector<int> dev_scans((size_t)nb, 0);
    int rc = jpeg_parallel(ctx, par, rest, files, order, dev_scans);
    if (rc) return rc;
    // progressive files the device has started: their coefficient arrays come to the host, which runs the remaining passes
    // (every progressive file's verdict is looked at here: JPG_REDO sends the file to the serial decoder)
    std::vector<int> prog, prog_all;
    for (int i : par) if (files[i].progressive) {
        prog_all.push_back(i);
        if ((size_t)dev_scans[i] < files[i].scans.size()) prog.push_back(i);
    }
    if (!prog_all.empty()) {
        need_coef();
        if (coef_rc) return coef_rc;
        const int16_t* d0 = reinterpret_cast<const int16_t*>(ctx->d_jpg);
        for (int i : prog)
            for (int c = 0; c < files[i].ncomp; c++) {
                const size_t off = (size_t)(ctx->h_jd[i].coef[c] - d0);
                I2S_HIP(hipMemcpyAsync(coef.data() + off, ctx->d_jpg + off * sizeof(int16_t), (size_t)files[i].c[c].bw * files[i].c[c].bh * 64 * sizeof(int16_t),
                                       hipMemcpyDeviceToHost, ctx->stream));
            }
        I2S_HIP(hipMemcpyAsync(ctx->h_jstatus, ctx->d_jstatus, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        const double t0 = now_ms();
        I2S_HIP(hipStreamSynchronize(ctx->stream));
        ctx->jpeg_ms[2] += (float)(now_ms() - t0);
        bool redo = false;
        for (int i : prog_all) {
            if (ctx->h_jstatus[i] == JPG_REDO) {
                // scans of this file wrote outside their bands: the whole file again, serially, in file order (jpeg_host_decode
                // starts from zeroed arrays and the upload replaces what the device left)
                late.push_back(i);
                prog.erase(std::remove(prog.begin(), prog.end(), i), prog.end());
                I2S_HIP(hipMemsetAsync(ctx->d_jstatus + i, 0, sizeof(int), ctx->stream));
                redo = true;
            } else if (ctx->h_jstatus[i] != JPG_OK) return jpeg_bad(ctx, order[i]);
        }
        (void)redo;
        const int pb = jpeg_host_finish(ctx, prog, files, order, dev_scans, coef);
        if (pb >= 0) return jpeg_bad(ctx, pb);
        rc = jpeg_host_upload(ctx, prog, files, coef);
        if (rc) return rc;
    }
    std::vector<uint8_t> bytes;
    rc = jpeg_lanes(ctx, lanes, files, jpeg, len, order, bytes);
    if (rc) return rc;
    if (host_job.joinable()) host_job.join();
    if (host_bad < 0 && !late.empty()) {
        need_coef();
        if (coef_rc) return coef_rc;
        host_bad = jpeg_host_decode(ctx, late, files, order, coef);
    }
    if (host_bad >= 0) return jpeg_bad(ctx, host_bad);
    rc = jpeg_host_upload(ctx, host, files, coef);
    if (rc) return rc;
    rc = jpeg_host_upload(ctx, late, files, coef);
    if (rc) return rc;
    I2S_HIP(hipMemcpyAsync(ctx->h_jstatus, ctx->d_jstatus, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    const double t0 = now_ms();
    I2S_HIP(hipStreamSynchronize(ctx->stream));                 // a corrupt file is reported before anything else runs
    ctx->jpeg_ms[2] += (float)(now_ms() - t0);
    for (int i = 0; i < nb; i++)
        if (ctx->h_jstatus[i] != JPG_OK) return jpeg_bad(ctx, order[i]);
    return I2S_OK;
}

extern "C" int i2s_jpeg_last_rounds(const i2s_ctx* ctx) { return ctx ? ctx->je_rounds : 0; }
extern "C" int i2s_jpeg_last_handed_back(const i2s_ctx* ctx) { return ctx ? ctx->je_handed_back : 0; }

extern "C" int i2s_jpeg_set_max_rounds(i2s_ctx* ctx, int rounds)
{
    if (!ctx || rounds < 1 || rounds > JE_MAX_ROUNDS - 16) return I2S_E_INVALID;
    ctx->je_max_rounds = rounds;
    return I2S_OK;
}

extern "C" int i2s_jpeg_last_timing(const i2s_ctx* ctx, float ms[4])
{
    if (!ctx || !ms) return I2S_E_INVALID;
    for (int i = 0; i < 4; i++) ms[i] = ctx->jpeg_ms[i];
    return I2S_OK;
}

extern "C" int i2s_detect_jpeg_batch(i2s_ctx* ctx, int B, const uint8_t* const* jpeg, const size_t* len, const i2s_xform* xf,
                                     const i2s_params* p, i2s_board* boards, i2s_result* full)
{
    try {
    if (!ctx || B < 0 || (B > 0 && (!jpeg || !len || !boards))) return I2S_E_INVALID;
    int rc = check_params(p);
    if (rc) return rc;
    const double t_call = now_ms();
    for (float& v : ctx->jpeg_ms) v = 0;
    struct Whole { i2s_ctx* c; double t; ~Whole() { c->jpeg_ms[3] = (float)(now_ms() - t); } } whole{ctx, t_call};
    // every file is checked (and its frame size noted) before anything runs; the parsed tables and scan lists are only kept
    // for the images of the pass in flight (a progressive file carries ~100 KB of Huffman tables)
    std::vector<int> fw(B), fh(B);
    std::vector<JpegFile> files;
    const bool one_pass = B <= ctx->max_batch;      // then the parsed files are kept; otherwise every pass parses its own again
    if (one_pass) files.resize(B);
    // parsed on the host threads (a parse walks the whole file for its markers); the first file in input order that fails decides
    std::vector<int> prc(B, I2S_OK);
    for (int i = 0; i < B; i++) if (!jpeg[i]) return I2S_E_INVALID;
    rc_parallel_for(B, [&](int i) {
        try {
            JpegFile scratch;
            JpegFile& f = one_pass ? files[i] : scratch;
            const int jr = jpg_parse(jpeg[i], len[i], &f);
            if (jr == JPG_BAD) prc[i] = I2S_E_INVALID;
            else if (jr == JPG_UNSUPPORTED) prc[i] = I2S_E_UNSUPPORTED;
            // sizes are checked before any workspace is sized from them (a hostile header may claim 65535 x 65535)
            else if (!xf && (f.X > ctx->max_w || f.Y > ctx->max_h)) prc[i] = I2S_E_TOO_LARGE;
            else if (xf && ((long long)f.X * f.Y > (1ll << 26) || f.X >= 32768 || f.Y >= 32768)) prc[i] = I2S_E_TOO_LARGE;
            fw[i] = f.X; fh[i] = f.Y;
        } catch (const std::exception&) { prc[i] = I2S_E_INVALID; }
    });
    for (int i = 0; i < B; i++) if (prc[i]) return prc[i];
    ctx->jpeg_ms[0] += (float)(now_ms() - t_call);
    I2S_HIP(hipSetDevice(ctx->device));
    // pass formation as in i2s_detect_batch_xf (by processed area when p->schedule is set)
    std::vector<int> order(B);
    for (int i = 0; i < B; i++) order[i] = i;
    if (p->schedule && B > ctx->max_batch) {
        auto area = [&](int i) {
            return xf ? (long long)(xf[i].crop[2] - xf[i].crop[0]) * (xf[i].crop[3] - xf[i].crop[1]) : (long long)fw[i] * fh[i];
        };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return area(a) < area(b); });
    }
    i2s_params pd = *p;
    pd.inputs_on_device = 1;
    pd.schedule = 0;
    float timing[5] = {0, 0, 0, 0, 0};
    JpegCoefHost& coef = ctx->h_coef;
    std::vector<i2s_board> pb(ctx->max_batch);
    std::vector<i2s_result> pf(full ? ctx->max_batch : 0);
    std::vector<i2s_xform> pxf(xf ? ctx->max_batch : 0);
    std::vector<const uint8_t*> ptr(ctx->max_batch);
    std::vector<int> pw(ctx->max_batch), ph(ctx->max_batch), ps(ctx->max_batch), pc(ctx->max_batch, 3);
    for (int first = 0; first < B; first += ctx->max_batch) {
        const int nb = B - first < ctx->max_batch ? B - first : ctx->max_batch;
        if (!one_pass) {
            const double t0 = now_ms();
            files.assign(nb, JpegFile());
            std::atomic<int> failed(0);
            rc_parallel_for(nb, [&](int i) {
                try { if (jpg_parse(jpeg[order[first + i]], len[order[first + i]], &files[i]) != JPG_OK) failed.store(1); }
                catch (const std::exception&) { failed.store(1); }
            });
            if (failed.load()) return I2S_E_INVALID;
            ctx->jpeg_ms[0] += (float)(now_ms() - t0);
        }
        // workspace layout of the pass: [coefficients of all images][component planes][RGB images]
        size_t ncoef = 0, nplane = 0, nrgb = 0;
        for (int i = 0; i < nb; i++) {
            const JpegFile& f = files[i];
            for (int c = 0; c < f.ncomp; c++) {
                ncoef += align256((size_t)f.c[c].bw * f.c[c].bh * 64 * sizeof(int16_t));
                nplane += align256((size_t)f.c[c].bw * f.c[c].bh * 64);
            }
            nrgb += align256((size_t)f.X * f.Y * 3);
        }
        const size_t need = ncoef + nplane + nrgb + 256;          // + slack: the classifier's dword loads may reach 3 bytes further
        if (need > ctx->jpg_bytes) {
            I2S_HIP(hipStreamSynchronize(ctx->stream));
            if (ctx->d_jpg) I2S_HIP(hipFree(ctx->d_jpg));
            ctx->d_jpg = nullptr; ctx->jpg_bytes = 0;
            I2S_HIP(hipMalloc(&ctx->d_jpg, need));
            ctx->jpg_bytes = need;
        }
        // layout first, then the entropy decoding of the pass's images (jpeg_entropy_pass below)
        size_t co = 0, po = ncoef, ro = ncoef + nplane;
        int wmax = 0, hmax = 0, max_blocks = 0;
        for (int i = 0; i < nb; i++) {
            const int k = order[first + i];
            const JpegFile& f = files[i];
            JpgDesc& J = ctx->h_jd[i];
            int blocks = 0;
            for (int c = 0; c < 3; c++) {
                J.coef[c] = nullptr; J.plane[c] = nullptr; J.bw[c] = J.bh[c] = J.dw[c] = J.dh[c] = J.nblocks[c] = 0;
            }
            for (int c = 0; c < f.ncomp; c++) {
                const size_t nblk = (size_t)f.c[c].bw * f.c[c].bh;
                J.coef[c] = reinterpret_cast<const int16_t*>(ctx->d_jpg + co);
                J.plane[c] = ctx->d_jpg + po;
                co += align256(nblk * 64 * sizeof(int16_t));
                po += align256(nblk * 64);
                J.bw[c] = f.c[c].bw; J.bh[c] = f.c[c].bh; J.dw[c] = f.c[c].dw; J.dh[c] = f.c[c].dh;
                J.nblocks[c] = (int)nblk;
                blocks += (int)nblk;
                for (int q = 0; q < 64; q++) J.q[c][q] = f.qc[c][q];
            }
            J.ncomp = f.ncomp; J.X = f.X; J.Y = f.Y; J.hs = f.c[0].h; J.vs = f.c[0].v;
            J.out = ctx->d_jpg + ro; J.out_stride = f.X * 3;
            ro += align256((size_t)f.X * f.Y * 3);
            ptr[i] = J.out; pw[i] = f.X; ph[i] = f.Y; ps[i] = f.X * 3;
            if (xf) pxf[i] = xf[k];
            wmax = f.X > wmax ? f.X : wmax; hmax = f.Y > hmax ? f.Y : hmax;
            max_blocks = blocks > max_blocks ? blocks : max_blocks;
        }
        rc = jpeg_entropy_pass(ctx, nb, files, jpeg, len, order.data() + first, p->jpeg_entropy_device, ncoef, coef);
        if (rc) return rc;
        I2S_HIP(hipMemcpyAsync(ctx->d_jd, ctx->h_jd, nb * sizeof(JpgDesc), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_jpeg_idct, dim3(cdiv(max_blocks, 64), nb), dim3(64), 0, ctx->stream, ctx->d_jd);
        hipLaunchKernelGGL(k_jpeg_rgb, dim3(cdiv(wmax, 64), cdiv(hmax, 4), nb), dim3(64, 4), 0, ctx->stream, ctx->d_jd);
        // the decoded images are device-resident sources of the ordinary path
        // The inner call sees pass-local image indices 0 .. nb - 1: with the caller's board sink in place it would write every
        // pass's records to sink[0 .. nb).  The sink is taken away for the call and the pass's records (still in d_boards) are
        // delivered afterwards to sink[input index], on the stream, before anything can overwrite d_boards.
        i2s_board* const sink = ctx->d_sink;
        ctx->d_sink = nullptr;
        rc = i2s_detect_batch_xf(ctx, nb, ptr.data(), pw.data(), ph.data(), ps.data(), pc.data(), xf ? pxf.data() : nullptr, &pd,
                                 pb.data(), full ? pf.data() : nullptr);
        ctx->d_sink = sink;
        if (rc) return rc;
        if (sink) {
            bool dense = true;
            for (int i = 1; i < nb; i++) dense &= order[first + i] == order[first] + i;
            if (dense) I2S_HIP(hipMemcpyAsync(sink + order[first], ctx->d_boards, (size_t)nb * sizeof(i2s_board), hipMemcpyDeviceToDevice, ctx->stream));
            else for (int i = 0; i < nb; i++)
                I2S_HIP(hipMemcpyAsync(sink + order[first + i], ctx->d_boards + i, sizeof(i2s_board), hipMemcpyDeviceToDevice, ctx->stream));
            I2S_HIP(hipStreamSynchronize(ctx->stream));
        }
        for (int i = 0; i < nb; i++) {
            boards[order[first + i]] = pb[i];
            if (full) full[order[first + i]] = pf[i];
        }
        for (int i = 0; i < 5; i++) timing[i] += ctx->timing[i];
    }
    for (int i = 0; i < 5; i++) ctx->timing[i] = timing[i];
    ctx->last_staged = 1;          // i2s_fetch_source: the decoded (and, if requested, transformed / enhanced) image
    return I2S_OK;
    } catch (const std::exception& e) {               // bad_alloc, a thread that could not be started ...: nothing crosses the C ABI
        snprintf(ctx->err, sizeof(ctx->err), "host failure while decoding JPEG data: %s", e.what());
        return I2S_E_INVALID;
    }
}
