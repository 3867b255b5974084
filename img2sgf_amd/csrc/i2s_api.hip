// C ABI of libi2s_hip.so (include/i2s.h): context, device workspace and the kernel pipeline.
// One context = one GPU = one HIP stream; a detect call streams the batch through the device in passes of
// `max_batch` images with a single synchronisation per pass.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <exception>
#include <thread>
#include <vector>

#include "i2s_types.h"
#include "comm_rccl.h"
#include "k_canny.h"
#include "k_erase_lines.h"
#include "k_filters.h"
#include "k_grid.h"
#include "jpeg_host.h"
#include "k_canny_rows.h"
#include "k_hough_circles.h"
#include "k_jpeg.h"
#include "k_jpeg_entropy.h"
#include "k_preprocess.h"

using namespace i2s;

namespace {
constexpr int NPLANES = I2S_PLANE__COUNT;    // 8 variant planes, removed, 9 maps
constexpr int HYST_MAX_PASSES = 4096;
constexpr int RAD_GX = 32;
}  // namespace

// host copy of JPEG coefficient arrays (progressive files): pinned, owned by the context, grown on demand
struct JpegCoefHost {
    int16_t* p = nullptr;
    size_t n = 0;
    int16_t* data() const { return p; }
    size_t size() const { return n; }
    int reserve(struct i2s_ctx* ctx, size_t count);
};

struct ProfPair { int seg; hipEvent_t start, stop; };

struct i2s_ctx {
    int device = 0, max_batch = 0, max_w = 0, max_h = 0;
    Geo geo{};
    hipStream_t stream = nullptr;
    hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
    uint8_t* d_planes = nullptr;
    uint8_t* d_src = nullptr;
    size_t src_slot = 0;
    ImgDesc* d_desc = nullptr;
    ImgDesc* h_desc = nullptr;
    int* d_flags = nullptr;      // [2][HYST_MAX_PASSES] tiles pass p queued for pass p + 1, per phase | [2] grid-barrier counters | [2] passes used (k_hysteresis_tail)
    int* h_flags = nullptr;      // [2] passes each phase needed (-1: not converged)
    unsigned* d_cent_list = nullptr;
    int* d_counts = nullptr;     // cent_count | est_count | vcount | overflow
    unsigned long long* d_est_keys = nullptr;
    float* d_vcirc = nullptr;
    unsigned long long* d_lsum = nullptr;   // [nb] luma sums for the contrast step
    int last_staged = 0;
    XfDesc* d_xf = nullptr;      // [max_batch] pre-transform descriptors (i2s_detect_batch_xf)
    XfDesc* h_xf = nullptr;
    uint8_t* d_raw = nullptr;    // untransformed host sources of one pass (grown on demand)
    size_t raw_bytes = 0;
    uint8_t* d_jpg = nullptr;    // JPEG decoding workspace of one pass: coefficients | component planes | RGB images (grown on demand)
    size_t jpg_bytes = 0;
    JpgDesc* d_jd = nullptr;     // [max_batch]
    JpgDesc* h_jd = nullptr;
    uint8_t* d_jh = nullptr;     // entropy decoding on the device: file bytes | Huffman tables | scans | images | status (grown on demand)
    size_t jh_bytes = 0;
    int* h_jstatus = nullptr;    // [max_batch] pinned
    int* d_jstatus = nullptr;    // [max_batch] verdicts of the device decoders, [max_batch] files the parallel decoder handed back (skip mask)
    uint8_t* d_je = nullptr;     // parallel entropy decoding (k_jpeg_entropy.h): bytes | tables | scans | segments | per-subsequence state (grown on demand)
    size_t je_bytes = 0;
    uint32_t* h_jflag = nullptr; // pinned
    uint8_t* h_jblob = nullptr;  // pinned: the destuffed entropy-coded bytes of a pass (grown on demand)
    size_t jblob_bytes = 0;
    JpegCoefHost h_coef;         // pinned: coefficient arrays of progressive files on their way to / from the host decoder
    int je_rounds = 0;           // rounds k_je_sync took in the last pass
    int je_max_rounds = 48;      // the files still iterating then are handed to the serial decoder (i2s_jpeg_set_max_rounds)
    int je_handed_back = 0;      // how many that were in the last pass
    float jpeg_ms[4] = {0, 0, 0, 0};     // last i2s_detect_jpeg_batch: parsing | entropy stage, host work | entropy stage, waiting for the device | whole call
    int* d_tl_cnt = nullptr;     // [nb][tiles] circles whose erase box touches the tile
    TlBox* d_tl_idx = nullptr;   // [nb][tiles][TL_CAP] box, plus centre and index of the circles that touch the tile
    int* d_weak = nullptr;       // 2 worklists (main Canny / HoughCircles' Cannys): [0] = count, then keys of tiles holding weak pixels
    int* d_chg = nullptr;        // hysteresis revisit queues: [2 (pass parity)][NMAP * nb * tiles] tile keys
    int* d_hmark = nullptr;      // [NMAP][nb][tiles] stamp of the latest pass a tile was queued for (hy_stamp + pass + 1; only ever grows)
    int hy_stamp = 0;            // advanced by HYST_MAX_PASSES + 2 per phase run
    int* d_colour = nullptr;     // [nb] k_grey: the 3-channel image holds a pixel whose channels differ (else it is treated as grey)
    int* d_mflags = nullptr;     // [nb][bands_y][bands_x] k_median57_bin: the band holds a pixel other than 0 / 255
    uint2* d_bin_ent = nullptr;
    int* d_bin_cnt = nullptr;
    int* d_lacc = nullptr;
    int lrow = 0;
    i2s_result* d_res = nullptr;
    i2s_board* d_boards = nullptr;
    i2s_board* h_boards = nullptr;
    // full records: what is in use of each is packed on the device (k_pack_results) and crosses the bus in ONE copy into pinned memory
    uint8_t* d_pack = nullptr; uint8_t* h_pack = nullptr; size_t pack_cap = 0;
    unsigned long long* d_pack_off = nullptr; unsigned long long* h_pack_off = nullptr;     // [nb] byte offset of image i's packed record
    i2s_board* d_sink = nullptr;    // i2s_set_board_sink: device array that also receives image i's record at [i]
    int* d_dbg_acc = nullptr;
    int debug = 0;
    bool hy_no_tail = false;        // a grid barrier of k_hysteresis_tail timed out once (a shared GPU): plain launches only from then on
    int hyst_k[2] = {1, 1};         // plain hysteresis launches per phase in front of the persistent tail: what the last call needed
    long long n_passes = 0, n_redone = 0;      // device passes run since i2s_create / of those, passes run again because a phase had not converged
    int hyst_used_max[2] = {0, 0};  // most hysteresis passes a phase has needed so far (plain launches + what the tail added)
    int last_nb = 0;
    HoughTrig last_trig{};
    float timing[5] = {0, 0, 0, 0, 0};
    int prof = 0;                   // i2s_set_profiling: per-kernel HIP events on the stream
    std::vector<ProfPair> prof_pairs;   // event pairs of the launches of the current pass (profiling)
    size_t prof_used = 0;
    float seg_ms[I2S_NSEG] = {};
    char err[256] = {0};
};

// segments of i2s_last_kernel_timing, in launch order
static const char* const kSegName[I2S_NSEG] = {
    "k_grey", "k_sobel_nms(main Canny)", "k_hysteresis(main Canny)", "k_median57_bin", "k_blur", "k_median57",
    "k_sobel_nms_rows(HoughCircles x7)", "k_hysteresis(HoughCircles)", "k_edge_bins", "k_vote_centres", "k_radius",
    "k_circles_final", "k_concat_circles+k_erase_lines+k_line_peaks", "k_grid"};
// Kernel launches of the detection pass.  With profiling on (i2s_set_profiling) a launch goes through hipExtLaunchKernelGGL, which
// stamps the kernel's own start and stop into a pair of events: i2s_last_kernel_timing then reports kernel DURATIONS per group, the
// quantity a rocprofv3 kernel trace reports -- rounds 1-3 recorded one event in front of every group, which also counted the
// dispatch gaps and the markers themselves (7 % on the blur+Canny stage: profiles/r04_roofline.md).
static bool prof_pair(i2s_ctx* ctx, int seg, hipEvent_t* s, hipEvent_t* e);
#define I2S_LAUNCH(seg, kernel, grid, block, ...)                                                                     \
    do {                                                                                                              \
        hipEvent_t ps_, pe_;                                                                                          \
        if (ctx->prof && prof_pair(ctx, seg, &ps_, &pe_))                                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, ps_, pe_, 0, __VA_ARGS__);                      \
        else hipLaunchKernelGGL(kernel, grid, block, 0, ctx->stream, __VA_ARGS__);                                    \
    } while (0)

#define I2S_HIP(call)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            snprintf(ctx->err, sizeof(ctx->err), "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                     __FILE__, __LINE__);                                                               \
            return I2S_E_HIP;                                                                           \
        }                                                                                               \
    } while (0)

static inline uint8_t* plane_ptr(i2s_ctx* ctx, int plane) { return ctx->d_planes + (size_t)plane * ctx->geo.nb * ctx->geo.slot; }
static inline int* cent_count(i2s_ctx* c) { return c->d_counts; }
static inline int* est_count(i2s_ctx* c) { return c->d_counts + (size_t)c->max_batch * NVAR; }
static inline int* vcount(i2s_ctx* c) { return c->d_counts + (size_t)2 * c->max_batch * NVAR; }
static inline int* overflow(i2s_ctx* c) { return c->d_counts + (size_t)3 * c->max_batch * NVAR; }
static inline size_t counts_bytes(i2s_ctx* c) { return ((size_t)3 * c->max_batch * NVAR + c->max_batch) * sizeof(int); }

extern "C" int i2s_abi_version(void) { return I2S_ABI_VERSION; }

extern "C" void i2s_default_params(i2s_params* p)
{
    memset(p, 0, sizeof(*p));
    p->canny_lo = 50; p->canny_hi = 200;
    p->hc_min_dist = 10.f; p->hc_param1 = 100; p->hc_param2 = 30; p->hc_min_radius = 1; p->hc_max_radius = 30;
    p->line_threshold = 0; p->black_threshold = 128;
    p->align_x = I2S_ALIGN_LEFT; p->align_y = I2S_ALIGN_TOP;
    p->min_grid_spacing = 10; p->big_space_ratio = 1.6; p->angle_tolerance_deg = 1.0;
    p->grey_shift = 15; p->gauss_kernel_mode = 0; p->houghlines_numangle_mode = 1; p->inputs_on_device = 0; p->schedule = 0; p->jpeg_entropy_device = 1;
    p->contrast = -1; p->brightness = -1;
}

// choose_threshold (img2sgf.py:606-613)
extern "C" int i2s_choose_threshold(int w, int h)
{
    const int x = w < h ? w : h;
    int t = (int)(x / 12.8 + 16);
    t = t < 20 ? 20 : (t > 200 ? 200 : t);
    return t;
}

extern "C" const char* i2s_strerror(int code)
{
    switch (code) {
        case I2S_OK: return "ok";
        case I2S_E_INVALID: return "invalid argument";
        case I2S_E_NO_DEVICE: return "no HIP device / device initialisation failed";
        case I2S_E_HIP: return "HIP runtime error";
        case I2S_E_TOO_LARGE: return "image larger than the context was created for";
        case I2S_E_UNSUPPORTED: return "parameter outside the supported envelope";
        default: return "unknown error";
    }
}

extern "C" const char* i2s_last_error(const i2s_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int i2s_device_arch(const i2s_ctx* ctx, char* buf, size_t cap)
{
    if (!ctx || !buf || cap == 0) return I2S_E_INVALID;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, ctx->device) != hipSuccess) return I2S_E_HIP;
    snprintf(buf, cap, "%s", prop.gcnArchName);
    return I2S_OK;
}

extern "C" void i2s_destroy(i2s_ctx* ctx)
{
    if (!ctx) return;
    void* dev[] = {ctx->d_planes, ctx->d_src, ctx->d_desc, ctx->d_flags, ctx->d_cent_list, ctx->d_counts, ctx->d_est_keys,
                   ctx->d_vcirc, ctx->d_lacc, ctx->d_res, ctx->d_boards, ctx->d_dbg_acc, ctx->d_bin_ent, ctx->d_bin_cnt, ctx->d_weak, ctx->d_chg, ctx->d_hmark, ctx->d_tl_cnt, ctx->d_tl_idx, ctx->d_mflags, ctx->d_colour, ctx->d_lsum, ctx->d_xf, ctx->d_raw, ctx->d_jpg, ctx->d_jd, ctx->d_jh, ctx->d_je, ctx->d_jstatus, ctx->d_pack, ctx->d_pack_off};
    for (void* q : dev) if (q) (void)hipFree(q);
    void* host[] = {ctx->h_desc, ctx->h_flags, ctx->h_boards, ctx->h_xf, ctx->h_jd, ctx->h_jstatus, ctx->h_jflag, ctx->h_jblob, ctx->h_coef.p, ctx->h_pack, ctx->h_pack_off};
    for (void* q : host) if (q) (void)hipHostFree(q);
    for (int i = 0; i < 5; i++) if (ctx->ev[i]) (void)hipEventDestroy(ctx->ev[i]);
    for (ProfPair& pp : ctx->prof_pairs) { (void)hipEventDestroy(pp.start); (void)hipEventDestroy(pp.stop); }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

static int create_impl(i2s_ctx* ctx)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || ctx->device >= ndev) return I2S_E_NO_DEVICE;
    I2S_HIP(hipSetDevice(ctx->device));
    I2S_HIP(hipStreamCreate(&ctx->stream));
    for (int i = 0; i < 5; i++) I2S_HIP(hipEventCreate(&ctx->ev[i]));
    Geo& g = ctx->geo;
    g.pitch = (ctx->max_w + 63) / 64 * 64;
    g.hmax = ctx->max_h;
    g.wmax = ctx->max_w;
    g.nb = ctx->max_batch;
    g.slot = (long long)g.pitch * g.hmax;
    g.bw = (ctx->max_w + EB - 1) / EB;
    g.bins = g.bw * ((ctx->max_h + EB - 1) / EB);
    g.tw = (ctx->max_w + CT_W - 1) / CT_W;
    g.tiles = g.tw * ((ctx->max_h + CT_H - 1) / CT_H);
    {
        const long long area = (long long)ctx->max_w * ctx->max_h;
        int f = (int)((area + (1 << 20) - 1) >> 20);
        f = f < 1 ? 1 : (f > CAP_SCALE_MAX ? CAP_SCALE_MAX : f);
        // accumulator maxima can be two orders of magnitude more numerous than circles (a ring of radius 5 under a 7x7 blur
        // yields a cloud of them): one entry per 8 pixels of the largest image, never less than the round-1 constant
        const long long cc = area / 8;
        g.cent_cap = (int)(cc < CENT_UNIT ? CENT_UNIT : (cc > (1 << 22) ? (1 << 22) : cc));
        g.est_cap = EST_UNIT * f; g.vcirc_cap = VCIRC_UNIT * f;
    }
    const size_t nb = ctx->max_batch;
    I2S_HIP(hipMalloc(&ctx->d_lsum, nb * sizeof(unsigned long long)));
    I2S_HIP(hipMalloc(&ctx->d_tl_cnt, nb * g.tiles * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_tl_idx, nb * g.tiles * TL_CAP * sizeof(TlBox)));
    I2S_HIP(hipMalloc(&ctx->d_weak, 2 * (nb * NMAP * g.tiles + 1) * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_chg, 2 * nb * NMAP * g.tiles * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_hmark, nb * NMAP * g.tiles * sizeof(int)));
    I2S_HIP(hipMemset(ctx->d_hmark, 0, nb * NMAP * g.tiles * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_mflags, nb * mb_bands_x(g.wmax) * mb_bands_y(g.hmax) * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_colour, nb * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_planes, (size_t)NPLANES * nb * g.slot + 256));
    ctx->src_slot = (size_t)ctx->max_w * 3 * ctx->max_h;
    I2S_HIP(hipMalloc(&ctx->d_src, nb * ctx->src_slot + 256));
    I2S_HIP(hipMalloc(&ctx->d_desc, nb * sizeof(ImgDesc)));
    I2S_HIP(hipHostMalloc(&ctx->h_desc, nb * sizeof(ImgDesc)));
    I2S_HIP(hipMalloc(&ctx->d_xf, nb * sizeof(XfDesc)));
    I2S_HIP(hipHostMalloc(&ctx->h_xf, nb * sizeof(XfDesc)));
    I2S_HIP(hipMalloc(&ctx->d_jd, nb * sizeof(JpgDesc)));
    I2S_HIP(hipHostMalloc(&ctx->h_jd, nb * sizeof(JpgDesc)));
    I2S_HIP(hipHostMalloc(&ctx->h_jstatus, nb * sizeof(int)));
    I2S_HIP(hipHostMalloc(&ctx->h_jflag, sizeof(uint32_t)));
    I2S_HIP(hipMalloc(&ctx->d_jstatus, 2 * nb * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_flags, (2 * HYST_MAX_PASSES + 4) * sizeof(int)));
    I2S_HIP(hipHostMalloc(&ctx->h_flags, 2 * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_cent_list, nb * NVAR * g.cent_cap * sizeof(unsigned)));
    I2S_HIP(hipMalloc(&ctx->d_counts, counts_bytes(ctx)));
    I2S_HIP(hipMalloc(&ctx->d_est_keys, nb * NVAR * g.est_cap * sizeof(unsigned long long)));
    I2S_HIP(hipMalloc(&ctx->d_vcirc, nb * NVAR * g.vcirc_cap * 3 * sizeof(float)));
    I2S_HIP(hipMalloc(&ctx->d_bin_cnt, nb * NVAR * g.bins * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_bin_ent, nb * NVAR * g.bins * EB_CAP * sizeof(uint2)));
    ctx->lrow = (2 * (ctx->max_w + ctx->max_h) + 1 + 15) / 16 * 16;
    I2S_HIP(hipMalloc(&ctx->d_lacc, nb * LROWS * ctx->lrow * sizeof(int)));
    I2S_HIP(hipMalloc(&ctx->d_res, nb * sizeof(i2s_result)));
    I2S_HIP(hipMalloc(&ctx->d_boards, nb * sizeof(i2s_board)));
    I2S_HIP(hipHostMalloc(&ctx->h_boards, nb * sizeof(i2s_board)));
    I2S_HIP(hipMemsetAsync(ctx->d_res, 0, nb * sizeof(i2s_result), ctx->stream));
    I2S_HIP(hipStreamSynchronize(ctx->stream));
    return I2S_OK;
}

extern "C" int i2s_create(i2s_ctx** out, int device_id, int max_batch, int max_w, int max_h)
{
    // (k_vote_centres packs "record index inside a pair of HoughCircles inputs | direction << 31" into one dword)
    static_assert(2ll * ((16384 + EB - 1) / EB) * ((16384 + EB - 1) / EB) * EB_CAP < (1ll << 31), "vote items: 31-bit record index");
    if (!out || device_id < 0 || max_batch < 1 || max_batch > 4096 || max_w < 1 || max_h < 1 || max_w > 16384 || max_h > 16384)
        return I2S_E_INVALID;
    i2s_ctx* ctx = new i2s_ctx();
    ctx->device = device_id; ctx->max_batch = max_batch; ctx->max_w = max_w; ctx->max_h = max_h;
    ctx->hy_no_tail = getenv("I2S_HYST_NO_TAIL") != nullptr;      // tests: the path a context takes after a grid-barrier timeout
    const int rc = create_impl(ctx);
    if (rc != I2S_OK) {
        if (rc == I2S_E_HIP) fprintf(stderr, "i2s_create: %s\n", ctx->err);
        i2s_destroy(ctx);
        *out = nullptr;
        return rc == I2S_E_HIP ? I2S_E_NO_DEVICE : rc;
    }
    *out = ctx;
    return I2S_OK;
}

extern "C" int i2s_set_debug(i2s_ctx* ctx, int on)
{
    if (!ctx) return I2S_E_INVALID;
    if (on && !ctx->d_dbg_acc)
        I2S_HIP(hipMalloc(&ctx->d_dbg_acc, (size_t)ctx->max_batch * NVAR * ctx->geo.slot * sizeof(int)));
    ctx->debug = on ? 1 : 0;
    return I2S_OK;
}

// 8-bit fixed-point Gaussian taps (OpenCV getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED).
static void gauss_taps(int k, double sigma, int mode, Taps* t)
{
    memset(t, 0, sizeof(*t));
    const double sig = sigma > 0 ? sigma : ((k - 1) * 0.5 - 1) * 0.3 + 0.8;
    const double scale2x = -0.125 / (sig * sig);
    const int n2 = (k - 1) / 2;
    double vals[8], sum = 0, kern[8];
    for (int i = 0, x = 1 - k; i < n2; i++, x += 2) { vals[i] = exp((double)(x * x) * scale2x); sum += vals[i]; }
    sum = sum * 2 + 1.0;
    const double mul = 1.0 / sum;
    for (int i = 0; i < n2; i++) { kern[i] = vals[i] * mul; kern[k - 1 - i] = kern[i]; }
    kern[n2] = mul;
    if (mode == 1) { for (int i = 0; i < k; i++) t->k[i] = (int)lrint(kern[i] * 256.0); return; }
    double err = 0; long s = 0;
    for (int i = 0; i < n2; i++) {
        const double adj = kern[i] * 256.0 + err;
        const int v0 = (int)lrint(adj);
        err = adj - (double)v0;
        t->k[i] = v0; t->k[k - 1 - i] = v0; s += 2 * v0;
    }
    t->k[n2] = (int)(256 - s);
}

// numangle + trig tables of the three HoughLines calls of find_lines (img2sgf.py:236-244; hough.cpp
// computeNumangle / createTrigTable; rho = 1 and theta = pi/180 narrowed to float by the C++ signature).
static int hough_trig(const i2s_params* p, HoughTrig* t)
{
    const double PI = 3.14159265358979323846;   // math.pi == CV_PI
    const double delta = PI / 180 * p->angle_tolerance_deg;
    const float theta = (float)(PI / 180.0);
    const double mins[3] = {PI / 2 - delta, 0.0, PI - delta};
    const double maxs[3] = {PI / 2 + delta, delta, PI};
    memset(t, 0, sizeof(*t));
    for (int c = 0; c < 3; c++) {
        int numangle;
        if (p->houghlines_numangle_mode == 1) numangle = (int)lrint((maxs[c] - mins[c]) / theta);
        else {
            numangle = (int)floor((maxs[c] - mins[c]) / theta) + 1;
            if (numangle > 1 && fabs(PI - (numangle - 1) * theta) < theta / 2) --numangle;
        }
        if (numangle < 0) numangle = 0;
        if (numangle > LANG) return I2S_E_UNSUPPORTED;
        t->n[c] = numangle;
        float ang = (float)mins[c];
        for (int n = 0; n < numangle; ang += theta, n++) {
            t->sin_[c][n] = (float)(sin((double)ang) * 1.0f);
            t->cos_[c][n] = (float)(cos((double)ang) * 1.0f);
        }
    }
    return I2S_OK;
}

// The used parts of the full records of a pass, packed: per image [the record up to `circles` + n_circles circles][n_circles kept flags,
// padded to 4][detected + board], at byte offset off[i] of `out` (run_pass computes the offsets from the board records it already has).
__global__ __launch_bounds__(256) void k_pack_results(const i2s_result* __restrict__ res, const i2s_board* __restrict__ boards,
                                                      const unsigned long long* __restrict__ off, uint8_t* __restrict__ out)
{
    const i2s_result* r = res + blockIdx.x;
    // n comes from the BOARD record: the host lays the offsets out and unpacks from its copy of that record (a capacity
    // overflow of the line peaks left the two counts different, and the record of n_circles circles overran its slot)
    const unsigned n = (unsigned)boards[blockIdx.x].n_circles;
    const unsigned* s32 = reinterpret_cast<const unsigned*>(r);
    unsigned* d32 = reinterpret_cast<unsigned*>(out + off[blockIdx.x]);
    const unsigned head = (unsigned)(offsetof(i2s_result, circles) / 4) + n * 3, kept = (n + 3) / 4;
    const unsigned tail = (unsigned)((sizeof(i2s_result) - offsetof(i2s_result, detected)) / 4);
    for (unsigned i = threadIdx.x; i < head; i += 256) d32[i] = s32[i];
    for (unsigned i = threadIdx.x; i < kept; i += 256) d32[head + i] = s32[offsetof(i2s_result, circle_kept) / 4 + i];
    for (unsigned i = threadIdx.x; i < tail; i += 256) d32[head + kept + i] = s32[offsetof(i2s_result, detected) / 4 + i];
}

static int check_params(const i2s_params* p)
{
    if (!p) return I2S_E_INVALID;
    if (p->hc_max_radius > 30 || p->hc_max_radius < 1 || p->hc_min_radius < 0 || p->hc_min_radius >= p->hc_max_radius)
        return I2S_E_UNSUPPORTED;
    if (p->hc_param2 < 0 || p->hc_param1 < 1 || p->canny_lo > p->canny_hi) return I2S_E_UNSUPPORTED;
    if (p->grey_shift != 15 && p->grey_shift != 14) return I2S_E_UNSUPPORTED;
    if (p->contrast > 100 || p->brightness > 100) return I2S_E_UNSUPPORTED;
    if (p->jpeg_entropy_device < 0 || p->jpeg_entropy_device > 2) return I2S_E_UNSUPPORTED;
    return I2S_OK;
}

static GridParams grid_params(const i2s_params* p)
{
    GridParams gp;
    gp.min_grid_spacing = p->min_grid_spacing; gp.big_space_ratio = p->big_space_ratio;
    gp.black_threshold = p->black_threshold; gp.align_x = p->align_x; gp.align_y = p->align_y; gp.pad = 0;
    return gp;
}

static inline int* worklist(i2s_ctx* ctx, int phase) { return ctx->d_weak + (size_t)phase * ((size_t)ctx->max_batch * NMAP * ctx->geo.tiles + 1); }

static bool prof_pair(i2s_ctx* ctx, int seg, hipEvent_t* s, hipEvent_t* e)
{
    if (ctx->prof_used == ctx->prof_pairs.size()) {
        ProfPair pp{seg, nullptr, nullptr};
        if (hipEventCreate(&pp.start) != hipSuccess) return false;
        if (hipEventCreate(&pp.stop) != hipSuccess) { (void)hipEventDestroy(pp.start); return false; }
        ctx->prof_pairs.push_back(pp);
    }
    ProfPair& pp = ctx->prof_pairs[ctx->prof_used++];
    pp.seg = seg;
    *s = pp.start; *e = pp.stop;
    return true;
}

// One phase of hysteresis (0: the main Canny's map, 1: HoughCircles' maps): hyst_k[phase] plain launches -- as many passes as the
// previous call needed -- and ONE persistent launch behind them that runs whatever is still necessary with grid-wide barriers and
// reports the number of passes the phase took (k_canny.h).  On diagrams that is 1 + 1 launches per phase.
static int run_hysteresis(i2s_ctx* ctx, int phase, int max_tiles)
{
    int* flags = ctx->d_flags + (size_t)phase * HYST_MAX_PASSES;
    int* counter = ctx->d_flags + 2 * HYST_MAX_PASSES + phase;
    int* info = ctx->d_flags + 2 * HYST_MAX_PASSES + 2 + phase;
    const int nblocks = max_tiles < HY_BLOCKS ? max_tiles : HY_BLOCKS;      // the worklist cannot be longer than max_tiles
    uint8_t* maps = plane_ptr(ctx, I2S_PLANE_CANNY_MAP);
    uint8_t* edges = phase == 0 ? plane_ptr(ctx, I2S_PLANE_EDGES) : (uint8_t*)nullptr;
    const int k = ctx->hyst_k[phase];
    const size_t queue_half = (size_t)ctx->max_batch * NMAP * ctx->geo.tiles;
    if (ctx->hy_stamp > 0x7fffffff - 4 * (HYST_MAX_PASSES + 2)) {         // once in ~250 000 calls: start the stamps over
        I2S_HIP(hipMemsetAsync(ctx->d_hmark, 0, queue_half * sizeof(int), ctx->stream));
        ctx->hy_stamp = 0;
    }
    const int stamp_base = ctx->hy_stamp;
    ctx->hy_stamp += HYST_MAX_PASSES + 2;
    const int seg = phase == 0 ? 2 : 7;            // i2s_last_kernel_timing: "k_hysteresis(main Canny)" / "k_hysteresis(HoughCircles)"
    for (int pass = 0; pass < k; pass++)
        I2S_LAUNCH(seg, k_hysteresis, dim3(nblocks), dim3(256), ctx->d_desc, ctx->geo, maps, edges, flags, pass,
                           worklist(ctx, phase), ctx->d_chg, queue_half, ctx->d_hmark, stamp_base);
    // the persistent tail runs whatever passes are still needed; after a barrier timeout (hy_no_tail) it is launched as ONE workgroup with
    // no pass budget, i.e. only to report which of the k plain launches reached the fixed point (-1: none, the host redoes with more)
    I2S_LAUNCH(seg, k_hysteresis_tail, dim3(ctx->hy_no_tail ? 1 : HY_TAIL_BLOCKS), dim3(256), ctx->d_desc, ctx->geo, maps, edges,
                       flags, k, ctx->hy_no_tail ? k : HYST_MAX_PASSES, worklist(ctx, phase), ctx->d_chg, queue_half, ctx->d_hmark, stamp_base,
                       counter, info);
    I2S_HIP(hipMemcpyAsync(&ctx->h_flags[phase], info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    return I2S_OK;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// Copy an image into a densely packed staging slot.  Contiguous sources (the usual numpy / torch case) go as ONE linear
// copy: hipMemcpy2DAsync falls back to a row-by-row path (~8.5 us per row, measured) when the row width is not a multiple
// of 4 bytes, which made real RGB scans with odd widths 10x slower end to end than their GPU time.
static hipError_t stage_image(uint8_t* dst, const uint8_t* src, size_t rowb, size_t stride, size_t rows, hipMemcpyKind kind,
                              hipStream_t st)
{
    if (stride == rowb) return hipMemcpyAsync(dst, src, rowb * rows, kind, st);
    return hipMemcpy2DAsync(dst, rowb, src, stride, rowb, rows, kind, st);
}

// One device pass over nb images whose descriptors are already in h_desc.
// Results of image i go to boards[dst[i]] / full[dst[i]].
static int run_pass(i2s_ctx* ctx, int nb, int wmax, int hmax, bool has_c1, bool has_c3, const i2s_params* p,
                    i2s_board* boards, i2s_result* full, const int* dst)
{
    bool dense = true;
    for (int i = 1; i < nb; i++) dense &= dst[i] == dst[0] + i;
    Geo& g = ctx->geo;
    g.nb = nb;
    hipStream_t st = ctx->stream;
    HoughTrig trig;
    int rc = hough_trig(p, &trig);
    if (rc) return rc;
    ctx->last_trig = trig;
    Taps t3, t5, t7;
    gauss_taps(3, 3, p->gauss_kernel_mode, &t3);
    gauss_taps(5, 5, p->gauss_kernel_mode, &t5);
    gauss_taps(7, 7, p->gauss_kernel_mode, &t7);
    const int hc_lo = p->hc_param1 / 2 > 1 ? p->hc_param1 / 2 : 1;
    const GridParams gp = grid_params(p);
    if (p->contrast >= 0 || p->brightness >= 0) {
        // ImageEnhance.Contrast / .Brightness (img2sgf.py:141-149) in place on the staged sources -- ONCE per pass, outside
        // the redo loop below (a redone pass must not enhance the image a second time)
        const float fc = p->contrast >= 0 ? (float)(102.0 / (101 - p->contrast) - 1) : 1.0f;
        const float fb = p->brightness >= 0 ? (float)(450.0 / (200 - p->brightness) - 2) : 1.0f;
        I2S_HIP(hipMemcpyAsync(ctx->d_desc, ctx->h_desc, nb * sizeof(ImgDesc), hipMemcpyHostToDevice, st));
        I2S_HIP(hipMemsetAsync(ctx->d_lsum, 0, nb * sizeof(unsigned long long), st));
        hipLaunchKernelGGL(k_luma_sum, dim3(cdiv(hmax, 8), nb), dim3(256), 0, st, ctx->d_desc, ctx->d_lsum);
        hipLaunchKernelGGL(k_enhance, dim3(cdiv(wmax * 3, 1024), hmax, nb), dim3(256), 0, st, ctx->d_desc, ctx->d_lsum, fc, fb);
    }

    // grey plane of every image: a single-channel source with dword-aligned rows IS its grey plane (cvtColor is the identity
    // there, img2sgf.py:153 on a grey input), everything else gets its slot of the plane array filled by k_grey
    bool need_grey = false;
    for (int i = 0; i < nb; i++) {
        ImgDesc& d = ctx->h_desc[i];
        // (width a multiple of 4 as well: the row kernels fetch whole dwords, which must not reach past the last pixel of a
        // buffer the library does not own)
        const bool alias = d.cn == 1 && ((uintptr_t)d.src & 3u) == 0 && (d.sstride & 3) == 0 && (d.w & 3) == 0;
        d.grey = alias ? d.src : plane_ptr(ctx, I2S_PLANE_GREY) + (size_t)i * g.slot;
        d.gpitch = alias ? d.sstride : g.pitch;
        need_grey |= !alias;
    }
    const int sum3 = t3.k[0] + t3.k[1] + t3.k[2], sum5 = t5.k[0] + t5.k[1] + t5.k[2] + t5.k[3] + t5.k[4];
    const int sum7 = t7.k[0] + t7.k[1] + t7.k[2] + t7.k[3] + t7.k[4] + t7.k[5] + t7.k[6];
    const bool float_blur = sum3 == 256 && sum5 == 256 && sum7 == 256;     // exactness condition of k_blur (see k_filters.h)
    BlurTaps bt;
    bt.c3 = (float)t3.k[1]; bt.a3 = (float)t3.k[0];
    bt.c5 = (float)t5.k[2]; bt.a5 = (float)t5.k[1]; bt.b5 = (float)t5.k[0];
    bt.c7 = (float)t7.k[3]; bt.a7 = (float)t7.k[2]; bt.b7 = (float)t7.k[1]; bt.d7 = (float)t7.k[0];

    for (;;) {
        ctx->prof_used = 0;                          // (a redone pass measures itself again)
        I2S_HIP(hipMemcpyAsync(ctx->d_desc, ctx->h_desc, nb * sizeof(ImgDesc), hipMemcpyHostToDevice, st));
        I2S_HIP(hipMemsetAsync(ctx->d_counts, 0, counts_bytes(ctx), st));
        I2S_HIP(hipMemsetAsync(ctx->d_flags, 0, (2 * HYST_MAX_PASSES + 4) * sizeof(int), st));
        I2S_HIP(hipMemsetAsync(worklist(ctx, 0), 0, sizeof(int), st));
        I2S_HIP(hipMemsetAsync(worklist(ctx, 1), 0, sizeof(int), st));
        I2S_HIP(hipMemsetAsync(ctx->d_lacc, 0, (size_t)nb * LROWS * ctx->lrow * sizeof(int), st));
        I2S_HIP(hipMemsetAsync(ctx->d_mflags, 0, (size_t)nb * mb_bands_x(g.wmax) * mb_bands_y(g.hmax) * sizeof(int), st));
        I2S_HIP(hipMemsetAsync(ctx->d_colour, 0, (size_t)nb * sizeof(int), st));
        uint8_t* grey = plane_ptr(ctx, I2S_PLANE_GREY);
        uint8_t* map0 = plane_ptr(ctx, I2S_PLANE_CANNY_MAP);
        uint8_t* edges = plane_ptr(ctx, I2S_PLANE_EDGES);
        const dim3 b64x4(64, 4), b256(256);
        // tile grids of this pass (1-D launches, XCD-aware tile order inside the kernels)
        const int rx = cdiv(wmax, 256), ry = cdiv(hmax, GREY_ROWS);    // k_grey / k_split_rgb: 256 x GREY_ROWS pixels per workgroup
        const int fx = cdiv(wmax, FT_W), fy = cdiv(hmax, FT_H);        // 64 x 32 tiles
        const int mx = cdiv(wmax, MT_W), my = cdiv(hmax, MT_H);        // 56 x 72 tiles
        const int ebx = cdiv(wmax, EBB_X * EB), eby = cdiv(hmax, EBB_Y * EB);      // 128 x 32 (4 x 1 edge bins)
        const int vx = cdiv(wmax, VT), vy = cdiv(hmax, VT);            // 126 x 126 accumulator cells
        const dim3 g_row((unsigned)rx * ry * nb), g_f((unsigned)fx * fy * nb), g_m((unsigned)mx * my * nb);

        I2S_HIP(hipEventRecord(ctx->ev[0], st));
        // -- kernel group 0 of i2s_last_kernel_timing
        if (need_grey) I2S_LAUNCH(0, k_grey, g_row, b64x4, ctx->d_desc, g, grey, p->grey_shift, ctx->d_colour, ctx->d_mflags, rx, ry);
        // Order of the two independent halves of the blur+Canny stage (round 4): the main Canny FIRST.  It is bound by its arithmetic and
        // does not care where the grey source comes from; k_blur is bound by its six plane stores, and those stream ~15 % faster when
        // the source it reads at the same time is already in the Infinity Cache than when HBM has to turn around between reads and
        // writes (profiles/r04_b_blur_experiments.txt: 1.92 -> 1.69 us per diagram with the source read by a kernel in front of it).
        // The main Canny and HoughCircles' internal Canny of the grey plane share everything but the high threshold when
        // their low thresholds coincide (the reference's 50 and 100 / 2): one kernel pass then writes both maps.
        // (3-channel images whose channels are equal everywhere -- greyscale scans opened as RGB -- take the single-channel kernel too:
        // k_grey has just found out which, d_colour)
        const bool rows_main = has_c1 || has_c3;
        const bool fused0 = rows_main && p->canny_lo == hc_lo;
        // -- kernel group 1 of i2s_last_kernel_timing
        const int cgx = cdiv(wmax, 1024), cgy = cdiv(hmax, CR_R);      // k_sobel_nms_rows: 4 wavefronts x 256 pixels, CR_R rows
        if (fused0)
            I2S_LAUNCH(1, (k_sobel_nms_rows<2, true>), dim3((unsigned)cgx * cgy * nb), b256, ctx->d_desc, g, grey, map0, edges, 0, hc_lo,
                               p->hc_param1, p->canny_hi, worklist(ctx, 1), worklist(ctx, 0), ctx->d_colour, (const int*)ctx->d_mflags, cgx, cgy);
        else if (rows_main)
            I2S_LAUNCH(1, (k_sobel_nms_rows<1, true>), dim3((unsigned)cgx * cgy * nb), b256, ctx->d_desc, g, grey, map0, edges, 0,
                               p->canny_lo, p->canny_hi, p->canny_hi, worklist(ctx, 1), worklist(ctx, 0), ctx->d_colour, (const int*)ctx->d_mflags, cgx, cgy);
        if (has_c3) {
            // coloured images (d_colour): their channels as three planes in the slots of the blur bank that are still free (median3, gauss3,
            // median5: consecutive), then the row kernel's colour mode (round 1's LDS-tile kernel took 680 us for the reference's nine colour
            // scans x 16, this pair 580; k_grey / k_split_rgb with 32 rows per workgroup instead of 4: 291 -> 166 us)
            static_assert(I2S_PLANE_GAUSS3 == I2S_PLANE_MEDIAN3 + 1 && I2S_PLANE_MEDIAN5 == I2S_PLANE_MEDIAN3 + 2, "three consecutive scratch planes");
            uint8_t* rgb = plane_ptr(ctx, I2S_PLANE_MEDIAN3);
            I2S_LAUNCH(1, k_split_rgb, g_row, b64x4, ctx->d_desc, g, rgb, ctx->d_colour, rx, ry);
            I2S_LAUNCH(1, (k_sobel_nms_rows<3, false>), dim3((unsigned)cgx * cgy * nb), b256, ctx->d_desc, g, rgb, map0, edges, 0, p->canny_lo,
                       p->canny_hi, p->canny_hi, worklist(ctx, 1), worklist(ctx, 0), ctx->d_colour, (const int*)ctx->d_mflags, cgx, cgy);
        }
        // -- kernel group 2 of i2s_last_kernel_timing
        rc = run_hysteresis(ctx, 0, fx * fy * nb);
        if (rc) return rc;
        // -- kernel group 3 of i2s_last_kernel_timing
        if (float_blur) {
            // the three Gaussians and the 3x3 median; bands of pure 0 / 255 pixels get their 5x5 / 7x7 medians here as well (majority
            // votes), the others are flagged for the bit-serial kernel
            const int bgx = cdiv(wmax, 1024), bgy = cdiv(hmax, BL_R);      // 4 wavefronts x 256 pixels, BL_R rows
            // -- kernel group 4 of i2s_last_kernel_timing  // (the k_median57_bin segment stays empty on this path)
            // speculative two-valued kernel (all six planes of the bands it finishes), then the general kernel on the bands it flagged.
            // The 16-bit sums of the first stay below 65536 until the last multiply-add (which saturates) only if every tap is positive
            bool bin_ok = true;
            for (int k = 0; k < 3; k++) bin_ok &= t3.k[k] > 0;
            for (int k = 0; k < 5; k++) bin_ok &= t5.k[k] > 0;
            for (int k = 0; k < 7; k++) bin_ok &= t7.k[k] > 0;
            if (bin_ok)
                I2S_LAUNCH(4, (k_blur<true>), dim3((unsigned)bgx * bgy * nb), b256, ctx->d_desc, g, plane_ptr(ctx, I2S_PLANE_MEDIAN3),
                                   plane_ptr(ctx, I2S_PLANE_GAUSS3), plane_ptr(ctx, I2S_PLANE_GAUSS5), plane_ptr(ctx, I2S_PLANE_GAUSS7),
                                   plane_ptr(ctx, I2S_PLANE_MEDIAN5), plane_ptr(ctx, I2S_PLANE_MEDIAN7), bt, ctx->d_mflags, bgx, bgy);
            else I2S_HIP(hipMemsetAsync(ctx->d_mflags, 0xff, (size_t)nb * mb_bands_x(g.wmax) * mb_bands_y(g.hmax) * sizeof(int), st));
            I2S_LAUNCH(4, (k_blur<false>), dim3((unsigned)bgx * bgy * nb), b256, ctx->d_desc, g, plane_ptr(ctx, I2S_PLANE_MEDIAN3),
                               plane_ptr(ctx, I2S_PLANE_GAUSS3), plane_ptr(ctx, I2S_PLANE_GAUSS5), plane_ptr(ctx, I2S_PLANE_GAUSS7),
                               plane_ptr(ctx, I2S_PLANE_MEDIAN5), plane_ptr(ctx, I2S_PLANE_MEDIAN7), bt, ctx->d_mflags, bgx, bgy);
        } else {
            // integer kernels (tap sets that do not sum to 256): the two-valued bands' 5x5 / 7x7 majority votes have a kernel of their own
            const int mbx = cdiv(wmax, 1024), mby = cdiv(hmax, MB_R);    // 4 wavefronts x 256 pixels, MB_R rows
            I2S_LAUNCH(4, k_median57_bin, dim3((unsigned)mbx * mby * nb), b256, ctx->d_desc, g, plane_ptr(ctx, I2S_PLANE_MEDIAN5),
                               plane_ptr(ctx, I2S_PLANE_MEDIAN7), ctx->d_mflags, mbx, mby);
            // -- kernel group 4 of i2s_last_kernel_timing
            I2S_LAUNCH(4, k_median3, g_f, b256, ctx->d_desc, g, plane_ptr(ctx, I2S_PLANE_MEDIAN3), fx, fy);
            I2S_LAUNCH(4, k_gauss357, g_f, b256, ctx->d_desc, g, plane_ptr(ctx, I2S_PLANE_GAUSS3),
                               plane_ptr(ctx, I2S_PLANE_GAUSS5), plane_ptr(ctx, I2S_PLANE_GAUSS7), t3, t5, t7, fx, fy);
        }
        // -- kernel group 5 of i2s_last_kernel_timing
        I2S_LAUNCH(5, k_median57, dim3((unsigned)cdiv(mx * my * nb, M_TPB)), b256, ctx->d_desc, g, plane_ptr(ctx, I2S_PLANE_MEDIAN5),
                           plane_ptr(ctx, I2S_PLANE_MEDIAN7), ctx->d_mflags, mx, my, mx * my * nb);
        I2S_HIP(hipEventRecord(ctx->ev[1], st));
        // -- kernel group 6 of i2s_last_kernel_timing
        const int v_first = fused0 ? 1 : 0;
        I2S_LAUNCH(6, (k_sobel_nms_rows<0, true>), dim3((unsigned)cgx * cgy * nb * (NVAR - v_first)), b256, ctx->d_desc, g, grey, map0,
                           (uint8_t*)nullptr, v_first, hc_lo, p->hc_param1, p->hc_param1, worklist(ctx, 1), worklist(ctx, 0), ctx->d_colour, (const int*)ctx->d_mflags, cgx, cgy);
        // -- kernel group 7 of i2s_last_kernel_timing
        rc = run_hysteresis(ctx, 1, fx * fy * nb * NVAR);
        if (rc) return rc;
        // -- kernel group 8 of i2s_last_kernel_timing
                I2S_LAUNCH(8, k_edge_bins, dim3((unsigned)ebx * eby * nb * NVAR), dim3(EBT), ctx->d_desc, g, grey, map0 + (size_t)nb * g.slot,
                           ctx->d_bin_ent, ctx->d_bin_cnt, ebx, eby);
        // -- kernel group 9 of i2s_last_kernel_timing
        // the reference's radius range (1 .. 30) gets the variant whose radius loop is unrolled
        {
            static_assert(NVAR % 2 == 0, "k_vote_centres pairs the HoughCircles inputs");
            const unsigned vgrid = (unsigned)vx * vy * nb * (NVAR / 2);
            // the reference's radius range (1 .. 30) gets the variant whose radius loop is unrolled
            if (p->hc_max_radius - p->hc_min_radius + 1 == 30)
                I2S_LAUNCH(9, (k_vote_centres<30>), dim3(vgrid), dim3(VPT), ctx->d_desc, g, ctx->d_bin_ent, ctx->d_bin_cnt,
                                   p->hc_min_radius, p->hc_max_radius, p->hc_param2, ctx->d_cent_list, cent_count(ctx),
                                   ctx->debug ? ctx->d_dbg_acc : (int*)nullptr, vx, vy);
            else
                I2S_LAUNCH(9, (k_vote_centres<0>), dim3(vgrid), dim3(VPT), ctx->d_desc, g, ctx->d_bin_ent, ctx->d_bin_cnt,
                                   p->hc_min_radius, p->hc_max_radius, p->hc_param2, ctx->d_cent_list, cent_count(ctx),
                                   ctx->debug ? ctx->d_dbg_acc : (int*)nullptr, vx, vy);
        }
        // -- kernel group 10 of i2s_last_kernel_timing
        I2S_LAUNCH(10, k_radius, dim3(RAD_GX, nb * NVAR), b256, ctx->d_desc, g, ctx->d_bin_ent, ctx->d_bin_cnt,
                           ctx->d_cent_list, cent_count(ctx), p->hc_min_radius, p->hc_max_radius, p->hc_param2,
                           ctx->d_est_keys, est_count(ctx));
        // -- kernel group 11 of i2s_last_kernel_timing
        if (g.est_cap <= EST_UNIT)
            I2S_LAUNCH(11, (k_circles_final<EST_UNIT, VCIRC_UNIT, true>), dim3(nb * NVAR), dim3(FIN_THREADS), g, ctx->d_est_keys,
                               est_count(ctx), cent_count(ctx), p->hc_min_dist, p->hc_min_radius, ctx->d_vcirc, vcount(ctx), overflow(ctx));
        else
            I2S_LAUNCH(11, (k_circles_final<EST_UNIT * CAP_SCALE_MAX, VCIRC_UNIT * CAP_SCALE_MAX, false>), dim3(nb * NVAR), dim3(FIN_THREADS), g, ctx->d_est_keys, est_count(ctx), cent_count(ctx), p->hc_min_dist, p->hc_min_radius, ctx->d_vcirc,
                               vcount(ctx), overflow(ctx));
        I2S_HIP(hipEventRecord(ctx->ev[2], st));
        // -- kernel group 12 of i2s_last_kernel_timing

        I2S_LAUNCH(12, k_concat_circles, dim3(nb), b256, ctx->d_desc, g, ctx->d_vcirc, vcount(ctx), overflow(ctx), ctx->d_res,
                           ctx->d_tl_cnt, ctx->d_tl_idx);
        const int ex = cdiv(wmax, ET_W), ey = cdiv(hmax, ET_H);         // 64 x 64 tiles
        I2S_LAUNCH(12, k_erase_lines, dim3((unsigned)ex * ey * nb), b256, ctx->d_desc, g, plane_ptr(ctx, I2S_PLANE_EDGES),
                           plane_ptr(ctx, I2S_PLANE_REMOVED), ctx->d_res, trig, ctx->d_lacc, ctx->lrow, ex, ey, ctx->d_tl_cnt, ctx->d_tl_idx);
        I2S_LAUNCH(12, k_line_peaks, dim3(nb), dim3(LP_THREADS), ctx->d_desc, ctx->d_lacc, ctx->lrow, trig, ctx->d_res);
        I2S_HIP(hipEventRecord(ctx->ev[3], st));
        // -- kernel group 13 of i2s_last_kernel_timing

        I2S_LAUNCH(13, k_grid, dim3(nb), dim3(GRID_THREADS), ctx->d_desc, g, gp, 1, ctx->d_res, ctx->d_boards);
        I2S_HIP(hipEventRecord(ctx->ev[4], st));
        // -- kernel group 14 of i2s_last_kernel_timing

        I2S_HIP(hipMemcpyAsync(ctx->h_boards, ctx->d_boards, nb * sizeof(i2s_board), hipMemcpyDeviceToHost, st));
        if (ctx->d_sink && dense)
            I2S_HIP(hipMemcpyAsync(ctx->d_sink + dst[0], ctx->d_boards, nb * sizeof(i2s_board), hipMemcpyDeviceToDevice, st));
        else if (ctx->d_sink) for (int i = 0; i < nb; i++)
            I2S_HIP(hipMemcpyAsync(ctx->d_sink + dst[i], ctx->d_boards + i, sizeof(i2s_board), hipMemcpyDeviceToDevice, st));
        I2S_HIP(hipStreamSynchronize(st));
        I2S_HIP(hipGetLastError());
        bool converged = true;
        ctx->n_passes++;
        for (int ph = 0; ph < 2; ph++) {
            const int used = ctx->h_flags[ph];                              // passes the phase needed, -1: budget or barrier timeout
            if (used > ctx->hyst_used_max[ph]) ctx->hyst_used_max[ph] = used;
            if (used < 0) {
                converged = false;
                if (used == -2) ctx->hy_no_tail = true;                     // no second second-long timeout on this context
                if (ctx->hyst_k[ph] >= HYST_MAX_PASSES) {
                    snprintf(ctx->err, sizeof(ctx->err), "Canny hysteresis did not converge in %d passes", HYST_MAX_PASSES);
                    return I2S_E_HIP;
                }
                ctx->hyst_k[ph] = ctx->hyst_k[ph] * 2 + 6 < HYST_MAX_PASSES ? ctx->hyst_k[ph] * 2 + 6 : HYST_MAX_PASSES;
            } else ctx->hyst_k[ph] = used < 1 ? 1 : (used > 64 ? 64 : used);
        }
        if (converged) break;
        // hysteresis had not reached its fixed point: redo this device pass with more plain launches
        ctx->n_redone++;
    }
    for (int i = 0; i < nb; i++) boards[dst[i]] = ctx->h_boards[i];
    if (full) {
        // The full record is 258 KB, nearly all of it the circle arrays' capacity: only what is in use crosses the bus -- the part
        // in front of the circles, n_circles circles, n_circles kept flags, the two boards behind them (the board record, already
        // on the host, says how many circles there are).  Array entries beyond the counts are left as the caller had them.
        // The used parts are packed on the device and come over in one copy into pinned memory (three copies per image
        // into the caller's pageable records serialised, 3 x nb calls per pass).
        constexpr size_t HEAD = offsetof(i2s_result, circles), TAIL = sizeof(i2s_result) - offsetof(i2s_result, detected);
        static_assert(HEAD % 4 == 0 && offsetof(i2s_result, circle_kept) % 4 == 0 && offsetof(i2s_result, detected) % 4 == 0 && TAIL % 4 == 0, "dword copies");
        if (!ctx->h_pack_off) {
            I2S_HIP(hipHostMalloc(&ctx->h_pack_off, (size_t)ctx->max_batch * sizeof(unsigned long long)));
            I2S_HIP(hipMalloc(&ctx->d_pack_off, (size_t)ctx->max_batch * sizeof(unsigned long long)));
        }
        size_t total = 0;
        for (int i = 0; i < nb; i++) {
            const size_t n = (size_t)ctx->h_boards[i].n_circles;
            ctx->h_pack_off[i] = total;
            total += HEAD + n * 12 + ((n + 3) & ~(size_t)3) + TAIL;
        }
        if (total > ctx->pack_cap) {
            if (ctx->d_pack) { (void)hipFree(ctx->d_pack); ctx->d_pack = nullptr; }
            if (ctx->h_pack) { (void)hipHostFree(ctx->h_pack); ctx->h_pack = nullptr; }
            ctx->pack_cap = 0;
            const size_t want = total + total / 4;
            I2S_HIP(hipMalloc(&ctx->d_pack, want));
            I2S_HIP(hipHostMalloc(&ctx->h_pack, want));
            ctx->pack_cap = want;
        }
        I2S_HIP(hipMemcpyAsync(ctx->d_pack_off, ctx->h_pack_off, nb * sizeof(unsigned long long), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(k_pack_results, dim3(nb), dim3(256), 0, st, ctx->d_res, ctx->d_boards, ctx->d_pack_off, ctx->d_pack);
        I2S_HIP(hipGetLastError());
        I2S_HIP(hipMemcpyAsync(ctx->h_pack, ctx->d_pack, total, hipMemcpyDeviceToHost, st));
        I2S_HIP(hipStreamSynchronize(st));
        for (int i = 0; i < nb; i++) {
            const size_t n = (size_t)ctx->h_boards[i].n_circles;
            const uint8_t* src = ctx->h_pack + ctx->h_pack_off[i];
            uint8_t* out = reinterpret_cast<uint8_t*>(full + dst[i]);
            memcpy(out, src, HEAD + n * 12);
            memcpy(out + offsetof(i2s_result, circle_kept), src + HEAD + n * 12, n);
            memcpy(out + offsetof(i2s_result, detected), src + HEAD + n * 12 + ((n + 3) & ~(size_t)3), TAIL);
        }
    }
    float ms;
    for (int i = 0; i < 4; i++) {
        I2S_HIP(hipEventElapsedTime(&ms, ctx->ev[i], ctx->ev[i + 1]));
        ctx->timing[i] += ms;
    }
    I2S_HIP(hipEventElapsedTime(&ms, ctx->ev[0], ctx->ev[4]));
    ctx->timing[4] += ms;
    if (ctx->prof) for (size_t i = 0; i < ctx->prof_used; i++) {
        I2S_HIP(hipEventElapsedTime(&ms, ctx->prof_pairs[i].start, ctx->prof_pairs[i].stop));
        ctx->seg_ms[ctx->prof_pairs[i].seg] += ms;
    }
    ctx->last_nb = nb;
    ctx->last_staged = (!p->inputs_on_device || p->contrast >= 0 || p->brightness >= 0) ? 1 : 0;
    return I2S_OK;
}

// Pillow Geometry.c: #define FIX(v) FLOOR((v) * 65536.0 + 0.5), FLOOR(v) = v >= 0 ? (int)v : (int)floor(v)
static int pil_fix(double v)
{
    v = v * 65536.0 + 0.5;
    return v >= 0.0 ? (int)v : (int)floor(v);
}

// affine_fixed()'s constants: the translation is moved to the pixel centre before it is converted
static void xform_fixed(const double a[6], XfDesc* x)
{
    x->a0 = pil_fix(a[0]); x->a1 = pil_fix(a[1]);
    x->a3 = pil_fix(a[3]); x->a4 = pil_fix(a[4]);
    x->a2 = pil_fix(a[2] + a[0] * 0.5 + a[1] * 0.5);
    x->a5 = pil_fix(a[5] + a[3] * 0.5 + a[4] * 0.5);
}

extern "C" int i2s_detect_batch_xf(i2s_ctx* ctx, int B, const uint8_t* const* img, const int* w, const int* h,
                                   const int* stride, const int* channels, const i2s_xform* xf, const i2s_params* p,
                                   i2s_board* boards, i2s_result* full)
{
    if (!ctx || B < 0 || (B > 0 && (!img || !w || !h || !stride || !channels || !boards))) return I2S_E_INVALID;
    int rc = check_params(p);
    if (rc) return rc;
    for (int i = 0; i < B; i++) {
        if (!img[i] || w[i] < 1 || h[i] < 1 || (channels[i] != 1 && channels[i] != 3) || stride[i] < w[i] * channels[i])
            return I2S_E_INVALID;
        if (xf) {
            const int32_t* c = xf[i].crop;
            if (c[2] <= c[0] || c[3] <= c[1]) return I2S_E_INVALID;
            if (w[i] >= 32768 || h[i] >= 32768) return I2S_E_UNSUPPORTED;      // Pillow leaves its fixed-point path there
            for (int k = 0; k < 6; k++) if (!(fabs(xf[i].affine[k]) < 32768.0)) return I2S_E_UNSUPPORTED;
            if ((long long)c[2] - c[0] > ctx->max_w || (long long)c[3] - c[1] > ctx->max_h) return I2S_E_TOO_LARGE;
        } else if (w[i] > ctx->max_w || h[i] > ctx->max_h) return I2S_E_TOO_LARGE;
    }
    I2S_HIP(hipSetDevice(ctx->device));
    for (int i = 0; i < 5; i++) ctx->timing[i] = 0;
    for (int i = 0; i < I2S_NSEG; i++) ctx->seg_ms[i] = 0;
    // pass formation: input order, or (ragged batches) ascending processed area so that similar sizes share a pass
    std::vector<int> order(B);
    for (int i = 0; i < B; i++) order[i] = i;
    if (p->schedule && B > ctx->max_batch) {
        auto area = [&](int i) {
            return xf ? (long long)(xf[i].crop[2] - xf[i].crop[0]) * (xf[i].crop[3] - xf[i].crop[1]) : (long long)w[i] * h[i];
        };
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return area(a) < area(b); });
    }
    for (int first = 0; first < B; first += ctx->max_batch) {
        const int nb = B - first < ctx->max_batch ? B - first : ctx->max_batch;
        int wmax = 0, hmax = 0;
        bool c1 = false, c3 = false;
        if (xf && !p->inputs_on_device) {
            // raw (untransformed) host sources of this pass go to their own staging buffer, grown on demand
            size_t need = 0;
            for (int i = 0; i < nb; i++) {
                const int k = order[first + i];
                need += ((size_t)w[k] * channels[k] * h[k] + 255) & ~(size_t)255;
            }
            if (need > ctx->raw_bytes) {
                I2S_HIP(hipStreamSynchronize(ctx->stream));
                if (ctx->d_raw) I2S_HIP(hipFree(ctx->d_raw));
                ctx->d_raw = nullptr; ctx->raw_bytes = 0;
                I2S_HIP(hipMalloc(&ctx->d_raw, need));
                ctx->raw_bytes = need;
            }
        }
        size_t raw_off = 0;
        for (int i = 0; i < nb; i++) {
            const int k = order[first + i];
            ImgDesc& d = ctx->h_desc[i];
            d.cn = channels[k]; d.gpitch = 0; d.grey = nullptr;
            const bool enhance = p->contrast >= 0 || p->brightness >= 0;
            uint8_t* slot = ctx->d_src + (size_t)i * ctx->src_slot;
            if (xf) {
                // crop_and_rotate_image (img2sgf.py:110-114) on the device: the staged source is the cropped region
                XfDesc& x = ctx->h_xf[i];
                const size_t rowb = (size_t)w[k] * channels[k];
                if (p->inputs_on_device) { x.src = img[k]; x.sstride = stride[k]; }
                else {
                    uint8_t* raw = ctx->d_raw + raw_off;
                    I2S_HIP(stage_image(raw, img[k], rowb, (size_t)stride[k], (size_t)h[k], hipMemcpyHostToDevice, ctx->stream));
                    x.src = raw; x.sstride = (int)rowb;
                    raw_off += (rowb * h[k] + 255) & ~(size_t)255;
                }
                x.sw = w[k]; x.sh = h[k];
                xform_fixed(xf[k].affine, &x);
                x.cl = xf[k].crop[0]; x.ct = xf[k].crop[1]; x.pad = 0;
                d.w = xf[k].crop[2] - xf[k].crop[0]; d.h = xf[k].crop[3] - xf[k].crop[1];
                d.src = slot; d.sstride = d.w * d.cn;
            } else {
                d.w = w[k]; d.h = h[k];
                if (p->inputs_on_device && !enhance) { d.src = img[k]; d.sstride = stride[k]; }
                else {
                    // staged copy (host inputs; device inputs that the contrast / brightness step will modify)
                    const size_t rowb = (size_t)w[k] * channels[k];
                    I2S_HIP(stage_image(slot, img[k], rowb, (size_t)stride[k], (size_t)h[k],
                                        p->inputs_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
                    d.src = slot; d.sstride = (int)rowb;
                }
            }
            d.line_thr = p->line_threshold > 0 ? p->line_threshold : i2s_choose_threshold(d.w, d.h);
            wmax = d.w > wmax ? d.w : wmax; hmax = d.h > hmax ? d.h : hmax;
            c1 |= channels[k] == 1; c3 |= channels[k] == 3;
        }
        if (xf) {
            I2S_HIP(hipMemcpyAsync(ctx->d_desc, ctx->h_desc, nb * sizeof(ImgDesc), hipMemcpyHostToDevice, ctx->stream));
            I2S_HIP(hipMemcpyAsync(ctx->d_xf, ctx->h_xf, nb * sizeof(XfDesc), hipMemcpyHostToDevice, ctx->stream));
            hipLaunchKernelGGL(k_rotate_crop, dim3(cdiv(wmax, 64), cdiv(hmax, 4), nb), dim3(64, 4), 0, ctx->stream, ctx->d_desc, ctx->d_xf);
        }
        rc = run_pass(ctx, nb, wmax, hmax, c1, c3, p, boards, full, order.data() + first);
        if (rc) return rc;
        if (xf) ctx->last_staged = 1;
    }
    return I2S_OK;
}

extern "C" int i2s_detect_batch(i2s_ctx* ctx, int B, const uint8_t* const* img, const int* w, const int* h,
                                const int* stride, const int* channels, const i2s_params* p,
                                i2s_board* boards, i2s_result* full)
{
    return i2s_detect_batch_xf(ctx, B, img, w, h, stride, channels, nullptr, p, boards, full);
}

#include "api_jpeg.h"     // JPEG input: i2s_jpeg_info, i2s_detect_jpeg_batch and the entropy stage of a pass

extern "C" int i2s_classify_batch(i2s_ctx* ctx, int first, int n, const i2s_params* p, i2s_board* boards, i2s_result* full)
{
    if (!ctx || !p || !boards || first < 0 || n < 1 || first + n > ctx->last_nb) return I2S_E_INVALID;
    I2S_HIP(hipSetDevice(ctx->device));
    const GridParams gp = grid_params(p);
    hipLaunchKernelGGL(k_grid, dim3(n), dim3(GRID_THREADS), 0, ctx->stream, ctx->d_desc + first, ctx->geo, gp, 0, ctx->d_res + first,
                       ctx->d_boards + first);
    I2S_HIP(hipMemcpyAsync(boards, ctx->d_boards + first, n * sizeof(i2s_board), hipMemcpyDeviceToHost, ctx->stream));
    if (full) I2S_HIP(hipMemcpyAsync(full, ctx->d_res + first, n * sizeof(i2s_result), hipMemcpyDeviceToHost, ctx->stream));
    I2S_HIP(hipStreamSynchronize(ctx->stream));
    I2S_HIP(hipGetLastError());
    return I2S_OK;
}

extern "C" int i2s_grid_from_lines(i2s_ctx* ctx, const uint8_t* grey, int w, int h, const float* circles, int n_circles,
                                   const float* hlines, int n_h, const float* vlines, int n_v, const i2s_params* p,
                                   i2s_board* board, i2s_result* full)
{
    if (!ctx || !grey || !p || !board || w < 1 || h < 1 || n_circles < 0 || n_h < 0 || n_v < 0) return I2S_E_INVALID;
    if (w > ctx->max_w || h > ctx->max_h) return I2S_E_TOO_LARGE;
    if (n_circles > I2S_MAX_CIRCLES || n_h > I2S_MAX_LINES || n_v > I2S_MAX_LINES) return I2S_E_UNSUPPORTED;
    I2S_HIP(hipSetDevice(ctx->device));
    Geo& g = ctx->geo;
    g.nb = 1;
    hipStream_t st = ctx->stream;
    i2s_result* hr = (i2s_result*)calloc(1, sizeof(i2s_result));
    if (!hr) return I2S_E_INVALID;
    hr->n_circles = n_circles; hr->n_hlines = n_h; hr->n_vlines = n_v;
    hr->line_threshold = p->line_threshold > 0 ? p->line_threshold : i2s_choose_threshold(w, h);
    if (n_circles) memcpy(hr->circles, circles, (size_t)n_circles * 3 * sizeof(float));
    if (n_h) memcpy(hr->hlines, hlines, (size_t)n_h * sizeof(float));
    if (n_v) memcpy(hr->vlines, vlines, (size_t)n_v * sizeof(float));
    ImgDesc& d = ctx->h_desc[0];
    d.src = nullptr; d.w = w; d.h = h; d.sstride = w; d.cn = 1; d.line_thr = hr->line_threshold;
    d.gpitch = g.pitch; d.grey = plane_ptr(ctx, I2S_PLANE_GREY);
    hipError_t e1 = hipMemcpyAsync(ctx->d_desc, ctx->h_desc, sizeof(ImgDesc), hipMemcpyHostToDevice, st);
    hipError_t e2 = hipMemcpyAsync(ctx->d_res, hr, sizeof(i2s_result), hipMemcpyHostToDevice, st);
    hipError_t e3 = hipMemcpy2DAsync(plane_ptr(ctx, I2S_PLANE_GREY), g.pitch, grey, w, w, h, hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k_grid, dim3(1), dim3(GRID_THREADS), 0, st, ctx->d_desc, g, grid_params(p), 1, ctx->d_res, ctx->d_boards);
    hipError_t e4 = hipMemcpyAsync(board, ctx->d_boards, sizeof(i2s_board), hipMemcpyDeviceToHost, st);
    hipError_t e5 = full ? hipMemcpyAsync(full, ctx->d_res, sizeof(i2s_result), hipMemcpyDeviceToHost, st) : hipSuccess;
    hipError_t e6 = hipStreamSynchronize(st);
    free(hr);
    I2S_HIP(e1); I2S_HIP(e2); I2S_HIP(e3); I2S_HIP(e4); I2S_HIP(e5); I2S_HIP(e6);
    I2S_HIP(hipGetLastError());
    ctx->last_nb = 1;
    return I2S_OK;
}

// validate_grid() (img2sgf.py:420-445) alone, on explicit cluster centres: what the reference's function takes -- float64 centres
// of any spacing and order, the circle list -- and its eight outputs in `out` (valid_grid, circle_kept / n_circles_kept, vsize,
// hsize, hcentres_complete, vcentres_complete, hspace, vspace; status says which axis failed and why).  A grid that does not
// validate (centres closer than min_grid_spacing, a single line, too wide a gap) is an ANSWER (valid_grid = 0, all circles kept,
// sizes 0), not an error.
extern "C" int i2s_validate_grid(i2s_ctx* ctx, const double* hcentres, int n_h, const double* vcentres, int n_v,
                                 const float* circles, int n_circles, const i2s_params* p, i2s_result* out)
{
    if (!ctx || !p || !out || n_h < 0 || n_v < 0 || n_circles < 0 || (n_h && !hcentres) || (n_v && !vcentres) || (n_circles && !circles))
        return I2S_E_INVALID;
    if (n_circles > I2S_MAX_CIRCLES || n_h > I2S_MAX_CENTRES || n_v > I2S_MAX_CENTRES) return I2S_E_UNSUPPORTED;
    I2S_HIP(hipSetDevice(ctx->device));
    Geo& g = ctx->geo;
    g.nb = 1;
    hipStream_t st = ctx->stream;
    i2s_result* hr = (i2s_result*)calloc(1, sizeof(i2s_result));
    if (!hr) return I2S_E_INVALID;
    hr->n_circles = n_circles; hr->n_hcentres = n_h; hr->n_vcentres = n_v;
    if (n_circles) memcpy(hr->circles, circles, (size_t)n_circles * 3 * sizeof(float));
    if (n_h) memcpy(hr->hcentres, hcentres, (size_t)n_h * sizeof(double));
    if (n_v) memcpy(hr->vcentres, vcentres, (size_t)n_v * sizeof(double));
    ImgDesc& d = ctx->h_desc[0];
    d.src = nullptr; d.w = 1; d.h = 1; d.sstride = 1; d.cn = 1; d.line_thr = 0;
    d.gpitch = g.pitch; d.grey = plane_ptr(ctx, I2S_PLANE_GREY);
    hipError_t e1 = hipMemcpyAsync(ctx->d_desc, ctx->h_desc, sizeof(ImgDesc), hipMemcpyHostToDevice, st);
    hipError_t e2 = hipMemcpyAsync(ctx->d_res, hr, sizeof(i2s_result), hipMemcpyHostToDevice, st);
    hipLaunchKernelGGL(k_grid, dim3(1), dim3(GRID_THREADS), 0, st, ctx->d_desc, g, grid_params(p), 2, ctx->d_res, ctx->d_boards);
    hipError_t e3 = hipMemcpyAsync(out, ctx->d_res, sizeof(i2s_result), hipMemcpyDeviceToHost, st);
    hipError_t e4 = hipStreamSynchronize(st);
    free(hr);
    I2S_HIP(e1); I2S_HIP(e2); I2S_HIP(e3); I2S_HIP(e4);
    I2S_HIP(hipGetLastError());
    ctx->last_nb = 0;                                   // no image behind this record: classify / fetch have nothing to refer to
    return I2S_OK;
}

// find_all_lines() (img2sgf.py:258-265) on an injected `circles_removed_image_np`: the three cv.HoughLines calls of
// find_lines (:236-244) for both directions, nothing else.
extern "C" int i2s_find_lines(i2s_ctx* ctx, const uint8_t* image, int w, int h, size_t stride, const i2s_params* p,
                              float* hlines, int* n_h, float* vlines, int* n_v)
{
    if (!ctx || !image || !p || !hlines || !vlines || !n_h || !n_v || w < 1 || h < 1 || stride < (size_t)w) return I2S_E_INVALID;
    if (w > ctx->max_w || h > ctx->max_h) return I2S_E_TOO_LARGE;
    I2S_HIP(hipSetDevice(ctx->device));
    Geo& g = ctx->geo;
    g.nb = 1;
    hipStream_t st = ctx->stream;
    HoughTrig trig;
    const int rc = hough_trig(p, &trig);
    if (rc) return rc;
    ctx->last_trig = trig;
    i2s_result* hr = (i2s_result*)calloc(1, sizeof(i2s_result));
    if (!hr) return I2S_E_INVALID;
    ImgDesc& d = ctx->h_desc[0];
    d.src = nullptr; d.w = w; d.h = h; d.sstride = w; d.cn = 1;
    d.line_thr = hr->line_threshold = p->line_threshold > 0 ? p->line_threshold : i2s_choose_threshold(w, h);
    d.gpitch = g.pitch; d.grey = plane_ptr(ctx, I2S_PLANE_GREY);
    const int fx = cdiv(w, ET_W), fy = cdiv(h, ET_H);
    hipError_t e[8];
    e[0] = hipMemcpyAsync(ctx->d_desc, ctx->h_desc, sizeof(ImgDesc), hipMemcpyHostToDevice, st);
    e[1] = hipMemcpyAsync(ctx->d_res, hr, sizeof(i2s_result), hipMemcpyHostToDevice, st);          // no circles: nothing is erased
    e[2] = hipMemcpy2DAsync(plane_ptr(ctx, I2S_PLANE_EDGES), g.pitch, image, stride, w, h, hipMemcpyHostToDevice, st);
    e[3] = hipMemsetAsync(ctx->d_lacc, 0, (size_t)LROWS * ctx->lrow * sizeof(int), st);
    hipLaunchKernelGGL(k_erase_lines, dim3((unsigned)fx * fy), dim3(256), 0, st, ctx->d_desc, g, plane_ptr(ctx, I2S_PLANE_EDGES),
                       plane_ptr(ctx, I2S_PLANE_REMOVED), ctx->d_res, trig, ctx->d_lacc, ctx->lrow, fx, fy, ctx->d_tl_cnt, ctx->d_tl_idx);
    hipLaunchKernelGGL(k_line_peaks, dim3(1), dim3(LP_THREADS), 0, st, ctx->d_desc, ctx->d_lacc, ctx->lrow, trig, ctx->d_res);
    e[4] = hipMemcpyAsync(hr, ctx->d_res, sizeof(i2s_result), hipMemcpyDeviceToHost, st);
    e[5] = hipStreamSynchronize(st);
    int out = I2S_OK;
    for (int i = 0; i < 6 && out == I2S_OK; i++)
        if (e[i] != hipSuccess) { snprintf(ctx->err, sizeof(ctx->err), "i2s_find_lines: %s", hipGetErrorString(e[i])); out = I2S_E_HIP; }
    if (out == I2S_OK && hr->status == I2S_ST_CAPACITY) out = I2S_E_UNSUPPORTED;                    // more than I2S_MAX_LINES peaks
    if (out == I2S_OK) {
        *n_h = hr->n_hlines; *n_v = hr->n_vlines;
        memcpy(hlines, hr->hlines, (size_t)hr->n_hlines * sizeof(float));
        memcpy(vlines, hr->vlines, (size_t)hr->n_vlines * sizeof(float));
    }
    free(hr);
    ctx->last_nb = 1;
    return out;
}

// Device -> host copy of `rows` rows of `rowb` bytes.  Dense on both sides: one linear copy.  Otherwise the whole pitched
// block comes over in one linear copy as well and the rows are compacted on the host (hipMemcpy2D's row-by-row path for widths
// that are not a multiple of 4 bytes costs ~8.5 us per row).
static int fetch_rows(i2s_ctx* ctx, uint8_t* dst, size_t dst_stride, const uint8_t* src, size_t src_pitch, size_t rowb, size_t rows)
{
    if (dst_stride == rowb && src_pitch == rowb) {
        I2S_HIP(hipMemcpyAsync(dst, src, rowb * rows, hipMemcpyDeviceToHost, ctx->stream));
        I2S_HIP(hipStreamSynchronize(ctx->stream));
        return I2S_OK;
    }
    const size_t bytes = src_pitch * (rows - 1) + rowb;
    uint8_t* tmp = (uint8_t*)malloc(bytes);
    if (!tmp) return I2S_E_INVALID;
    hipError_t e = hipMemcpyAsync(tmp, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) for (size_t y = 0; y < rows; y++) memcpy(dst + y * dst_stride, tmp + y * src_pitch, rowb);
    free(tmp);
    I2S_HIP(e);
    return I2S_OK;
}

extern "C" int i2s_fetch_plane(i2s_ctx* ctx, int index, int plane_id, uint8_t* dst, size_t dst_stride)
{
    if (!ctx || !dst || index < 0 || index >= ctx->last_nb || plane_id < 0 || plane_id >= NPLANES) return I2S_E_INVALID;
    const ImgDesc& d = ctx->h_desc[index];
    if (dst_stride < (size_t)d.w) return I2S_E_INVALID;
    if (plane_id == I2S_PLANE_GREY) return fetch_rows(ctx, dst, dst_stride, d.grey, (size_t)d.gpitch, (size_t)d.w, (size_t)d.h);
    const uint8_t* src = plane_ptr(ctx, plane_id) + (size_t)index * ctx->geo.slot;
    return fetch_rows(ctx, dst, dst_stride, src, ctx->geo.pitch, (size_t)d.w, (size_t)d.h);
}

extern "C" int i2s_fetch_source(i2s_ctx* ctx, int index, uint8_t* dst, size_t dst_stride)
{
    if (!ctx || !dst || index < 0 || index >= ctx->last_nb || !ctx->last_staged) return I2S_E_INVALID;
    const ImgDesc& d = ctx->h_desc[index];
    const size_t rowb = (size_t)d.w * d.cn;
    if (dst_stride < rowb) return I2S_E_INVALID;
    return fetch_rows(ctx, dst, dst_stride, d.src, (size_t)d.sstride, rowb, (size_t)d.h);
}

extern "C" int i2s_last_timing(const i2s_ctx* ctx, float ms[5])
{
    if (!ctx || !ms) return I2S_E_INVALID;
    for (int i = 0; i < 5; i++) ms[i] = ctx->timing[i];
    return I2S_OK;
}

extern "C" int i2s_set_profiling(i2s_ctx* ctx, int on)
{
    if (!ctx) return I2S_E_INVALID;
    I2S_HIP(hipSetDevice(ctx->device));
    ctx->prof = on ? 1 : 0;
    return I2S_OK;
}

extern "C" int i2s_blur_band_stats(i2s_ctx* ctx, int* flagged, int* total)
{
    if (!ctx || !flagged || !total || ctx->last_nb <= 0) return I2S_E_INVALID;
    I2S_HIP(hipSetDevice(ctx->device));
    const int nbx = mb_bands_x(ctx->geo.wmax), nby = mb_bands_y(ctx->geo.hmax);
    std::vector<int> f((size_t)ctx->last_nb * nbx * nby);
    I2S_HIP(hipMemcpy(f.data(), ctx->d_mflags, f.size() * sizeof(int), hipMemcpyDeviceToHost));
    int nf = 0, nt = 0;
    for (int b = 0; b < ctx->last_nb; b++) {
        const int bx = mb_bands_x(ctx->h_desc[b].w), by = mb_bands_y(ctx->h_desc[b].h);
        for (int y = 0; y < by; y++)
            for (int x = 0; x < bx; x++) { nt++; nf += f[((size_t)b * nby + y) * nbx + x] != 0; }
    }
    *flagged = nf; *total = nt;
    return I2S_OK;
}

extern "C" int i2s_hysteresis_stats(const i2s_ctx* ctx, long long* passes, long long* redone, int used_max[2])
{
    if (!ctx || !passes || !redone || !used_max) return I2S_E_INVALID;
    *passes = ctx->n_passes; *redone = ctx->n_redone;
    used_max[0] = ctx->hyst_used_max[0]; used_max[1] = ctx->hyst_used_max[1];
    return I2S_OK;
}

extern "C" int i2s_last_kernel_timing(const i2s_ctx* ctx, float ms[I2S_NSEG])
{
    if (!ctx || !ms || !ctx->prof) return I2S_E_INVALID;
    for (int i = 0; i < I2S_NSEG; i++) ms[i] = ctx->seg_ms[i];
    return I2S_OK;
}

extern "C" const char* i2s_kernel_timing_name(int i) { return i >= 0 && i < I2S_NSEG ? kSegName[i] : ""; }

extern "C" int i2s_fetch_circle_acc(i2s_ctx* ctx, int index, int variant, int32_t* dst)
{
    if (!ctx || !dst || !ctx->d_dbg_acc || index < 0 || index >= ctx->last_nb || variant < 0 || variant >= NVAR) return I2S_E_INVALID;
    const ImgDesc& d = ctx->h_desc[index];
    const int* src = ctx->d_dbg_acc + ((size_t)index * NVAR + variant) * ctx->geo.hmax * ctx->geo.pitch;
    I2S_HIP(hipMemcpy2DAsync(dst, (size_t)d.w * 4, src, (size_t)ctx->geo.pitch * 4, (size_t)d.w * 4, d.h, hipMemcpyDeviceToHost, ctx->stream));
    I2S_HIP(hipStreamSynchronize(ctx->stream));
    return I2S_OK;
}

extern "C" int i2s_fetch_line_acc(i2s_ctx* ctx, int index, int32_t* dst, size_t cap, int* numrho, int* nangles)
{
    if (!ctx || !dst || index < 0 || index >= ctx->last_nb) return I2S_E_INVALID;
    const ImgDesc& d = ctx->h_desc[index];
    const int nr = 2 * (d.w + d.h) + 1;
    if (cap < (size_t)LROWS * nr) return I2S_E_INVALID;
    const int* src = ctx->d_lacc + (size_t)index * LROWS * ctx->lrow;
    I2S_HIP(hipMemcpy2DAsync(dst, (size_t)nr * 4, src, (size_t)ctx->lrow * 4, (size_t)nr * 4, LROWS, hipMemcpyDeviceToHost, ctx->stream));
    I2S_HIP(hipStreamSynchronize(ctx->stream));
    if (numrho) *numrho = nr;
    if (nangles) { nangles[0] = ctx->last_trig.n[0]; nangles[1] = ctx->last_trig.n[1]; nangles[2] = ctx->last_trig.n[2]; }
    return I2S_OK;
}

// ---- multi-GPU: all-gather of the board records over RCCL (no reference counterpart) ---------------------------------

extern "C" int i2s_set_board_sink(i2s_ctx* ctx, i2s_board* d_sink)
{
    if (!ctx) return I2S_E_INVALID;
    ctx->d_sink = d_sink;
    return I2S_OK;
}

// text of the last failure of a call that has no communicator to hang it on (i2s_comm_unique_id, i2s_comm_create), per thread
static thread_local char g_comm_err[256] = "no error";

extern "C" const char* i2s_comm_last_error(const i2s_comm* comm) { return comm ? comm->err : g_comm_err; }

extern "C" void i2s_comm_destroy(i2s_comm* comm)
{
    if (!comm) return;
    (void)hipSetDevice(comm->device);
    RcclApi* api = rccl_api();
    if (comm->comm && api) (void)api->CommDestroy(comm->comm);
    if (comm->d_all) (void)hipFree(comm->d_all);
    delete comm;
}

static RcclApi* rccl_api_or_error()
{
    RcclApi* api = rccl_api();
    if (!api) snprintf(g_comm_err, sizeof(g_comm_err), "%s", rccl_open_error());
    return api;
}

extern "C" int i2s_comm_unique_id(uint8_t id[I2S_COMM_ID_BYTES])
{
    if (!id) return I2S_E_INVALID;
    RcclApi* api = rccl_api_or_error();
    if (!api) return I2S_E_NO_DEVICE;
    RcclId u;
    memset(&u, 0, sizeof(u));
    const int r = api->GetUniqueId(&u);
    if (r != 0) { snprintf(g_comm_err, sizeof(g_comm_err), "ncclGetUniqueId failed: %s", api->GetErrorString(r)); return I2S_E_HIP; }
    memcpy(id, u.internal, I2S_COMM_ID_BYTES);
    return I2S_OK;
}

extern "C" int i2s_comm_create(i2s_comm** out, int device_id, const uint8_t id[I2S_COMM_ID_BYTES], int world, int rank, int records_per_rank)
{
    if (!out || !id || device_id < 0 || world < 1 || rank < 0 || rank >= world || records_per_rank < 1 ||
        (long long)world * records_per_rank > (1ll << 24))
        return I2S_E_INVALID;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id >= ndev) { snprintf(g_comm_err, sizeof(g_comm_err), "no HIP device %d", device_id); return I2S_E_NO_DEVICE; }
    RcclApi* api = rccl_api_or_error();
    if (!api) return I2S_E_NO_DEVICE;
    if (hipSetDevice(device_id) != hipSuccess) { snprintf(g_comm_err, sizeof(g_comm_err), "hipSetDevice(%d) failed", device_id); return I2S_E_NO_DEVICE; }
    i2s_comm* c = new i2s_comm();
    c->device = device_id; c->world = world; c->rank = rank; c->cap = records_per_rank;
    // The gather buffer first, but a rank whose allocation fails STILL enters ncclCommInitRank: the call is collective, and a rank
    // that returned early would leave its peers waiting in it for ever.  The failure is reported after the rendezvous.
    const size_t bytes = (size_t)world * records_per_rank * sizeof(i2s_board);
    hipError_t he = hipMalloc(&c->d_all, bytes);
    if (he == hipSuccess) he = hipMemset(c->d_all, 0, bytes);
    RcclId u;
    memcpy(u.internal, id, I2S_COMM_ID_BYTES);
    const int r = api->CommInitRank(&c->comm, world, u, rank);       // collective: every rank of the job calls it
    if (r != 0) {
        snprintf(g_comm_err, sizeof(g_comm_err), "ncclCommInitRank failed: %s", api->GetErrorString(r));
        c->comm = nullptr;
        i2s_comm_destroy(c);
        return I2S_E_HIP;
    }
    if (he != hipSuccess) {
        snprintf(g_comm_err, sizeof(g_comm_err), "gather buffer of %zu bytes: %s", bytes, hipGetErrorString(he));
        i2s_comm_destroy(c);
        return I2S_E_HIP;
    }
    *out = c;
    return I2S_OK;
}

extern "C" i2s_board* i2s_comm_shard(i2s_comm* comm) { return comm ? comm->d_all + (size_t)comm->rank * comm->cap : nullptr; }
extern "C" i2s_board* i2s_comm_all(i2s_comm* comm) { return comm ? comm->d_all : nullptr; }

extern "C" int i2s_allgather_boards(i2s_ctx* ctx, i2s_comm* comm, const i2s_board* d_boards, int n_local, i2s_board* d_all, i2s_board* h_all)
{
    if (!ctx || !comm || n_local < 0 || n_local > comm->cap || ctx->device != comm->device) return I2S_E_INVALID;
    RcclApi* api = rccl_api();
    if (!api) return I2S_E_NO_DEVICE;
    I2S_HIP(hipSetDevice(ctx->device));
    i2s_board* shard = i2s_comm_shard(comm);
    if (!d_boards) d_boards = shard;
    if (!d_all) d_all = comm->d_all;
    // every rank sends comm->cap records (shards differ by at most one image).  Records handed in from elsewhere are first copied
    // into the own shard -- the caller's array need only hold n_local of them -- and the unused tail of the shard is zeroed.
    if (d_boards != shard) {
        if (n_local > 0) I2S_HIP(hipMemcpyAsync(shard, d_boards, (size_t)n_local * sizeof(i2s_board), hipMemcpyDeviceToDevice, ctx->stream));
        d_boards = shard;
    }
    if (n_local < comm->cap)
        I2S_HIP(hipMemsetAsync(shard + n_local, 0, (size_t)(comm->cap - n_local) * sizeof(i2s_board), ctx->stream));
    // in place (sendbuff == recvbuff + rank * count) when the result goes to the communicator's own buffer
    const int r = api->AllGather(d_boards, d_all, (size_t)comm->cap * sizeof(i2s_board), 1 /* ncclUint8 */, comm->comm, ctx->stream);
    if (r != 0) {
        snprintf(ctx->err, sizeof(ctx->err), "ncclAllGather failed: %s", api->GetErrorString(r));
        snprintf(comm->err, sizeof(comm->err), "%s", ctx->err);
        return I2S_E_HIP;
    }
    if (h_all) I2S_HIP(hipMemcpyAsync(h_all, d_all, (size_t)comm->world * comm->cap * sizeof(i2s_board), hipMemcpyDeviceToHost, ctx->stream));
    I2S_HIP(hipStreamSynchronize(ctx->stream));
    return I2S_OK;
}
