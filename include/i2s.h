/*
 * i2s.h -- C ABI of the MI355X board-detection library (libi2s_hip.so).
 *
 * The reference (hanysz/img2sgf) has no FFI: its hot path is wired through Python module
 * globals and direct cv2 calls.  This header is the boundary a maintainer binds with ctypes
 * (INTEGRATION.md shows the stub).  Each entry point names the reference interface it
 * replaces (file:line in img2sgf.py):
 *
 *   i2s_detect_batch      process_image() 117-204 + find_grid() 546-576 for a batch of images:
 *                         cv.cvtColor 153, cv.Canny 162, cv.medianBlur 174, cv.GaussianBlur 175,
 *                         10x cv.HoughCircles 180, rectangle/circle erase 191-198,
 *                         find_all_lines/find_lines 230-265 (3x cv.HoughLines),
 *                         cluster_lines 295-332, validate_grid 420-445, identify_board 497-543.
 *   i2s_classify_batch    apply_black_thresh() 762-766 -> identify_board() 497-543 only
 *                         (re-classify cached detections with a new black threshold / alignment).
 *   i2s_grid_from_lines   find_grid() 546-576 with injected circles and Hough-line rho lists
 *                         (what find_grid sees after find_lines 230-255 returned).
 *   i2s_find_lines        find_all_lines() 258-265 / find_lines() 230-255 on an injected circles_removed image.
 *   i2s_choose_threshold  choose_threshold() 606-613.
 *   i2s_detect_batch_xf   the same, preceded on the device by crop_and_rotate_image() 110-114
 *                         (PIL Image.rotate(NEAREST, fillcolor white, center) + Image.crop).
 *   i2s_detect_jpeg_batch the same from JPEG bytes: Image.open(...).convert("RGB") 651 for Huffman JPEGs, decoded on
 *                         the device (marker parsing on the host; entropy decoding, IDCT, upsampling, colour on the device).
 *   i2s_fetch_source      input_image_np 150 after the on-device rotate / crop (110-114) and contrast / brightness
 *                         (141-149) steps, if enabled.
 *   i2s_comm_* / i2s_allgather_boards / i2s_set_board_sink
 *                         no reference counterpart (the reference is one process): BASELINE.json's multi-GPU row -- batches
 *                         shard by image index across the GPUs of a node, one process per GPU, and the 384-byte board
 *                         records (what to_SGF 781-810 consumes) are all-gathered device to device over RCCL / xGMI.
 *   i2s_fetch_plane       the numpy images the GUI draws: grey_image_np 153,
 *                         edge_detected_image_np 162, the blur bank 171-175,
 *                         circles_removed_image_np 169-198.
 *
 * Conventions: plain pointers and sizes, no C++ types; all functions return 0 (I2S_OK) or a
 * negative I2S_E_* code and never throw; caller owns every buffer passed in; the context
 * owns all device memory; a context is single-caller (the reference is single-threaded) and
 * calls are synchronous on return.  There is NO CPU fallback: i2s_create fails with
 * I2S_E_NO_DEVICE when no gfx950 GPU is visible.
 */
#ifndef I2S_H
#define I2S_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define I2S_ABI_VERSION 4

#define I2S_BOARD_SIZE 19      /* img2sgf.py:43 */
#define I2S_NSLOTS 10          /* blur-bank slots, img2sgf.py:171-175 */
#define I2S_MAX_CIRCLES 16384  /* concatenated circles per image (all 10 slots) */
#define I2S_MAX_LINES 1024     /* Hough-line peaks per direction */
#define I2S_MAX_CENTRES 1024   /* cluster centres / completed grid lines per direction: = I2S_MAX_LINES (a cluster holds at least one
                                  line), so the reference's unbounded lists of centres are never cut short on their own account */

/* return codes */
enum {
    I2S_OK = 0,
    I2S_E_INVALID = -1,      /* bad argument */
    I2S_E_NO_DEVICE = -2,    /* no MI355X visible / HIP runtime error at create */
    I2S_E_HIP = -3,          /* HIP runtime error during a call (see i2s_last_error) */
    I2S_E_TOO_LARGE = -4,    /* image larger than the context was created for */
    I2S_E_UNSUPPORTED = -5   /* parameter outside the supported envelope */
};

/* per-image status (i2s_board.status / i2s_result.status); mirrors the reference's log lines */
enum {
    I2S_ST_BOARD_READY = 0,        /* board_ready = True (img2sgf.py:574) */
    I2S_ST_H_NO_LINES = 1,         /* "No grid lines found at all!" 340 (horizontal axis) */
    I2S_ST_H_ONE_LINE = 2,         /* "Only found one grid line" 344 */
    I2S_ST_H_TOO_CLOSE = 3,        /* "Grid lines are too close together" 351 */
    I2S_ST_H_TOO_WIDE = 4,         /* "Distance between edges of grid is ..." 371 */
    I2S_ST_V_NO_LINES = 5,
    I2S_ST_V_ONE_LINE = 6,
    I2S_ST_V_TOO_CLOSE = 7,
    I2S_ST_V_TOO_WIDE = 8,
    I2S_ST_TOO_MANY_VLINES = 9,    /* hsize > 19, "Too many vertical lines!" 569 */
    I2S_ST_TOO_MANY_HLINES = 10,   /* vsize > 19, "Too many horizontal lines!" 571 */
    I2S_ST_CAPACITY = 100          /* a capacity overflowed: results invalid.  The reference's lists are unbounded; here: I2S_MAX_CIRCLES
                                      concatenated circles and I2S_MAX_LINES peaks per direction in the record; per HoughCircles call
                                      2048 circles and 4096 supported estimates PER STARTED MEGAPIXEL of the context's max_w x max_h
                                      (up to 4x) and one accumulator maximum per 8 pixels: larger contexts have more room */
};

/* board cell values: BoardStates, img2sgf.py:82-83 */
enum { I2S_EMPTY = 0, I2S_BLACK = 1, I2S_WHITE = 2, I2S_STONE = 3 };
/* Alignment, img2sgf.py:86-87 */
enum { I2S_ALIGN_TOP = 0, I2S_ALIGN_BOTTOM = 1, I2S_ALIGN_LEFT = 2, I2S_ALIGN_RIGHT = 3 };

/* plane ids for i2s_fetch_plane */
enum {
    I2S_PLANE_GREY = 0,      /* grey_image_np */
    I2S_PLANE_EDGES = 1,     /* edge_detected_image_np (0/255) */
    I2S_PLANE_MEDIAN3 = 2, I2S_PLANE_GAUSS3 = 3,
    I2S_PLANE_MEDIAN5 = 4, I2S_PLANE_GAUSS5 = 5,
    I2S_PLANE_MEDIAN7 = 6, I2S_PLANE_GAUSS7 = 7,
    I2S_PLANE_REMOVED = 8,   /* circles_removed_image_np */
    I2S_PLANE_CANNY_MAP = 9, /* +m: Canny map m after hysteresis (2 = edge); m=0 main, 1..8 HoughCircles inputs */
    I2S_PLANE__COUNT = 18
};

typedef struct i2s_ctx i2s_ctx;

/* Parameters = the reference's constants (img2sgf.py:43-57) and hard-wired call arguments.
 * i2s_default_params() fills in the reference's values. */
typedef struct i2s_params {
    int32_t canny_lo, canny_hi;        /* 50, 200            img2sgf.py:47-48, 162.  cv.Canny's other two arguments are NOT
                                          parameters: apertureSize = 3 and L2gradient = false (the L1 norm) are the only flavour
                                          implemented.  The reference passes sobel.get() / gradient.get() == 2 (:164-165), but the
                                          widgets behind them are hidden (:1142-1182) and stay at 3 / L1; the GUI adapter refuses
                                          any other value (img2sgf_amd/gui_adapter.py) instead of ignoring it. */
    float   hc_min_dist;               /* 10                 :180 */
    int32_t hc_param1, hc_param2;      /* 100, 30            :180 */
    int32_t hc_min_radius, hc_max_radius; /* 1, 30           :180  (max_radius <= 30 supported) */
    int32_t line_threshold;            /* 0 = choose_threshold(w,h) per image (:638); else fixed */
    int32_t black_threshold;           /* 128                :45 */
    int32_t align_x, align_y;          /* I2S_ALIGN_LEFT, I2S_ALIGN_TOP  :627 */
    double  min_grid_spacing;          /* 10                 :54 */
    double  big_space_ratio;           /* 1.6                :55 */
    double  angle_tolerance_deg;       /* 1.0                :52 */
    /* OpenCV-version switches.  The reference pins no OpenCV version (it only logs cv.__version__,
     * img2sgf.py:1246); the defaults restate OpenCV 4.3 .. 4.5.1, the releases of the year it was written. */
    int32_t grey_shift;                /* 15 (4.x, default) | 14 (3.x) */
    int32_t gauss_kernel_mode;         /* 0 = error-diffused taps summing to 256 (>= 4.3, default) | 1 = plain rounding (4.0 - 4.2) */
    int32_t houghlines_numangle_mode;  /* 1 = cvRound(range / theta) angles (<= 4.5.1, default) | 0 = floor(..) + 1 with the pi-wrap fix (>= 4.5.2) */
    int32_t inputs_on_device;          /* 1: img[] are device pointers, read IN PLACE (no copy): a single-channel image with 4-byte
                                          aligned rows and a width that is a multiple of 4 also serves as its own grey plane: it must stay valid and unchanged until
                                          the next detect call if i2s_classify_batch / i2s_fetch_plane(GREY) are used on it;
                                          0: host pointers (staged into the context) */
    /* Pillow pre-processing on the device (img2sgf.py:141-149): contrast / brightness slider values 0..100, the
     * reference's defaults are 70 / 50.  -1 (default) = off: img[] already is `input_image_np` (:150). */
    int32_t contrast, brightness;
    /* Ragged batches: != 0 lets a call with more images than one device pass holds form its passes over the
     * images sorted by area, so that a pass's tile grids (sized for its largest image) are not mostly empty; results are
     * returned in input order.  The "last pass" the fetch / classify calls refer to is then the pass of the largest images. */
    int32_t schedule;
    /* i2s_detect_jpeg_batch, where the Huffman decoding runs.  1 (default): sequential files, and the scans of progressive files
     * in front of their first refinement pass (DC and AC first passes), on the device, parallel inside every scan
     * (csrc/k_jpeg_entropy.h); the remaining passes of progressive files afterwards on host threads, and files whose
     * entropy-coded data hold anything but stuffed FF00 bytes and RSTn markers wholly on host threads.  0: every file on host
     * threads (round 1's path).  2: sequential files as 1, progressive files and whatever the parallel decoder does not take on
     * the device too, one lane per file (slow: DESIGN.md 7a).  The three give the same coefficients, bit for bit, and refuse the
     * same files. */
    int32_t jpeg_entropy_device;
} i2s_params;

/* Compact per-image record: what the SGF writer needs (to_SGF 781-810) and what ranks
 * all-gather (384 bytes). board[i][j]: i = column (x), j = row (y), exactly full_board[i,j]. */
typedef struct i2s_board {
    uint8_t board[I2S_BOARD_SIZE][I2S_BOARD_SIZE];
    uint8_t status;          /* I2S_ST_* */
    uint8_t side_to_move;    /* 1 = black, 2 = white (img2sgf.py:89, 529-534); 0 if no board */
    uint8_t hsize, vsize;    /* detected grid size */
    uint8_t found_grid, valid_grid;
    uint8_t pad0;
    uint16_t n_black, n_white;
    uint16_t n_circles;      /* concatenated circle count */
    uint16_t line_threshold; /* Hough-lines threshold used */
    uint8_t pad[8];
} i2s_board;

/* Full per-image record: every value the reference leaves in its globals for the GUI.  Array entries beyond their counts
 * (circles[n_circles ..], circle_kept[n_circles ..], hlines[n_hlines ..] ...) are unspecified: a detect call copies only the
 * used part of the circle arrays to the host (the record's capacity is 258 KB, a diagram uses about 45). */
typedef struct i2s_result {
    int32_t status;
    int32_t line_threshold;
    int32_t found_grid, valid_grid, board_ready;
    int32_t hsize, vsize;
    int32_t n_circles;                      /* `circles` after the HoughCircles loop (:186) */
    int32_t n_per_slot[I2S_NSLOTS];         /* circles contributed by each blur-bank slot */
    int32_t n_circles_kept;                 /* after validate_grid's radius filter (:443) */
    int32_t n_hlines, n_vlines;             /* find_all_lines (:258-265) */
    int32_t n_hcentres, n_vcentres;         /* cluster_lines (:295-332) */
    int32_t n_hcomplete, n_vcomplete;       /* hcentres_complete / vcentres_complete */
    int32_t n_stones, n_black, n_white, side_to_move;
    int32_t pad0;
    double hspace, vspace;                  /* :437-438 */
    double hcentres[I2S_MAX_CENTRES], vcentres[I2S_MAX_CENTRES];
    double hcentres_complete[I2S_MAX_CENTRES], vcentres_complete[I2S_MAX_CENTRES];
    double brightness[I2S_BOARD_SIZE * I2S_BOARD_SIZE];   /* stone_brightnesses (:508-514) */
    float hlines[I2S_MAX_LINES], vlines[I2S_MAX_LINES];   /* rho, in find_lines' output order */
    float circles[I2S_MAX_CIRCLES][3];      /* x, y, r */
    uint8_t circle_kept[I2S_MAX_CIRCLES];   /* 1 if the circle survives the radius filter */
    uint8_t detected[I2S_BOARD_SIZE][I2S_BOARD_SIZE];     /* detected_board[hsize][vsize] */
    uint8_t board[I2S_BOARD_SIZE][I2S_BOARD_SIZE];        /* full_board */
    uint8_t pad1[2];
} i2s_result;

int  i2s_abi_version(void);
void i2s_default_params(i2s_params* p);
int  i2s_choose_threshold(int w, int h);
const char* i2s_strerror(int code);
const char* i2s_last_error(const i2s_ctx* ctx);
/* The architecture name of the device the context runs on ("gfx950:sramecc+:xnack-" on an MI355X), NUL-terminated into buf[cap];
 * returns I2S_OK, I2S_E_INVALID for a null argument / cap == 0.  No reference counterpart: it is how a caller (and the GPU test
 * suite) tells a context on the hardware from any other build of this ABI -- the CPU emulation used by tests/emu answers "emulated". */
int  i2s_device_arch(const i2s_ctx* ctx, char* buf, size_t cap);

/* device_id >= 0. max_batch = images processed per device pass (workspace is sized for it);
 * detect_batch accepts any B and loops over passes. */
int  i2s_create(i2s_ctx** out, int device_id, int max_batch, int max_w, int max_h);
void i2s_destroy(i2s_ctx* ctx);

/* img[b]: row-major, channels interleaved, channels[b] in {1,3}; stride[b] in bytes.
 * boards: [B] (required). full: [B] or NULL. */
int  i2s_detect_batch(i2s_ctx* ctx, int B, const uint8_t* const* img,
                      const int* w, const int* h, const int* stride, const int* channels,
                      const i2s_params* p, i2s_board* boards, i2s_result* full);

/* Pre-transform of one source image, applied on the device before everything else: crop_and_rotate_image() 110-114, i.e.
 * Pillow's Image.rotate(angle, NEAREST, fillcolor="white", center=c) followed by Image.crop(box).
 *   affine[6]: the inverse (output -> input) matrix Image.rotate() hands to Image.transform(AFFINE)
 *              (img2sgf_amd.preprocess.rotate_matrix computes it exactly as Pillow's Python code does);
 *   crop[4]:   (left, upper, right, lower) in the rotated image, which has the size of the source; right > left,
 *              lower > upper; the part of the box outside the image reads 0 as with Image.crop().
 * The detection then runs on the (right-left) x (lower-upper) region, which must fit the context's max_w x max_h. */
typedef struct i2s_xform {
    double  affine[6];
    int32_t crop[4];
} i2s_xform;

/* i2s_detect_batch with a per-image pre-transform xf[B] (NULL = none, identical to i2s_detect_batch).  w/h/stride describe
 * the SOURCE images; the contrast / brightness step of p (141-149), if enabled, follows the transform as in the reference. */
int  i2s_detect_batch_xf(i2s_ctx* ctx, int B, const uint8_t* const* img,
                         const int* w, const int* h, const int* stride, const int* channels,
                         const i2s_xform* xf, const i2s_params* p, i2s_board* boards, i2s_result* full);

/* JPEG input: Image.open(path).convert("RGB") (img2sgf.py:651) for 8-bit Huffman JPEGs, sequential or
 * progressive (grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0, any scan script), bit-exact with Pillow / libjpeg-turbo: marker parsing
 * on the host, entropy decoding on the device for sequential files (i2s_params.jpeg_entropy_device), dequantisation +
 * inverse DCT + chroma upsampling + colour conversion on the device, then the ordinary path (xf
 * and the contrast / brightness step of p apply to the decoded image).  I2S_E_UNSUPPORTED for any other flavour (arithmetic
 * coding, CMYK, RGB-coded, 12-bit, lossless): nothing is approximated, decode those elsewhere and use i2s_detect_batch(_xf).
 * i2s_jpeg_info reports the frame size (and 1 or 3 components) or the same error codes without decoding. */
int  i2s_jpeg_info(const uint8_t* data, size_t len, int* w, int* h, int* channels);
int  i2s_detect_jpeg_batch(i2s_ctx* ctx, int B, const uint8_t* const* jpeg, const size_t* len,
                           const i2s_xform* xf, const i2s_params* p, i2s_board* boards, i2s_result* full);
/* Rounds the parallel entropy decoder's iteration took in the last pass of the last i2s_detect_jpeg_batch call (0: it did not
 * run -- mode 0, or no sequential file in the pass).  Diagnostic. */
int  i2s_jpeg_last_rounds(const i2s_ctx* ctx);
/* The iteration's limit (default 48 rounds: the reference's scans need 15-17, 1024x1024 diagrams 7-9).  The FILES that still have
 * work scheduled then -- streams of identical blocks, e.g. a blank page (one 1024-bit subsequence per round: 129 rounds for
 * 1024 x 1024), or crafted ones -- are decoded by the serial decoder instead (host threads in mode 1, lanes in mode 2); the other
 * files of the pass are finished and stay on the device.  i2s_jpeg_last_handed_back: how many files of the last pass that was. */
int  i2s_jpeg_last_handed_back(const i2s_ctx* ctx);
int  i2s_jpeg_set_max_rounds(i2s_ctx* ctx, int rounds);
/* Host-side wall times of the last i2s_detect_jpeg_batch call, ms: [0] marker parsing, [1] the entropy stage's host work
 * (removing the byte stuffing and building the records, or the Huffman decoding itself on host threads), [2] waiting for the
 * device inside the entropy stage, [3] the whole call.  Diagnostic. */
int  i2s_jpeg_last_timing(const i2s_ctx* ctx, float ms[4]);

/* Re-run the stone classifier on the images of the LAST pass of the last detect call
 * (first..first+n) with p->black_threshold / p->align_*; circles, lines and grid are reused. */
int  i2s_classify_batch(i2s_ctx* ctx, int first, int n, const i2s_params* p,
                        i2s_board* boards, i2s_result* full);

/* find_grid() on injected data for ONE image: grey (host, w*h, stride w), circles [n][3],
 * hlines/vlines rho lists exactly as find_lines returns them. */
int  i2s_grid_from_lines(i2s_ctx* ctx, const uint8_t* grey, int w, int h,
                         const float* circles, int n_circles,
                         const float* hlines, int n_h, const float* vlines, int n_v,
                         const i2s_params* p, i2s_board* board, i2s_result* full);

/* validate_grid() (img2sgf.py:420-445) alone: the reference's arguments -- cluster centres as float64 (any spacing, any order:
 * whatever get_cluster_centres or a caller produced), the circle list -- and its eight outputs in `out`: valid_grid, circle_kept[] /
 * n_circles_kept (newcircles, :443), vsize, hsize, hcentres_complete, vcentres_complete, hspace, vspace.  A grid that fails
 * (the reference returns [False, circles, 0, 0, None, None, None, None], :424/:430) comes back as valid_grid = 0, every circle
 * kept, sizes 0 and `status` naming the axis and the reason (I2S_ST_H_TOO_CLOSE ...): an answer, never an error code.
 * n_h, n_v <= I2S_MAX_CENTRES, n_circles <= I2S_MAX_CIRCLES (I2S_E_UNSUPPORTED beyond).  min_grid_spacing / big_space_ratio of p apply. */
int  i2s_validate_grid(i2s_ctx* ctx, const double* hcentres, int n_h, const double* vcentres, int n_v,
                       const float* circles, int n_circles, const i2s_params* p, i2s_result* out);

/* find_all_lines() (img2sgf.py:258-265) alone: the three cv.HoughLines calls of find_lines (:236-244) for both directions
 * on an injected `circles_removed_image_np` (host, h rows of w bytes, `stride` bytes apart; any non-zero byte votes).
 * hlines / vlines: caller buffers of I2S_MAX_LINES floats, rho in find_lines' output order.  I2S_E_UNSUPPORTED if more
 * than I2S_MAX_LINES peaks exist in a direction. */
int  i2s_find_lines(i2s_ctx* ctx, const uint8_t* image, int w, int h, size_t stride, const i2s_params* p,
                    float* hlines, int* n_h, float* vlines, int* n_v);

/* Copy one plane of image `index` of the last pass to host memory (dst: h rows of w bytes). */
int  i2s_fetch_plane(i2s_ctx* ctx, int index, int plane_id, uint8_t* dst, size_t dst_stride);

/* Copy the (enhanced) source image of image `index` of the last pass: h rows of w * channels bytes.  Only available
 * when the pass staged its inputs (host inputs, or contrast/brightness enabled). */
int  i2s_fetch_source(i2s_ctx* ctx, int index, uint8_t* dst, size_t dst_stride);

/* Stage timing of the last detect call, milliseconds measured with HIP events on the context's
 * stream: [0] blur+Canny (grey, 3 medians, 3 Gaussians, main Canny incl. hysteresis),
 * [1] Hough circles x8 (their Sobel/Canny, votes, centres, radii, min-dist), [2] erase + Hough lines,
 * [3] grid + classifier, [4] total. */
int  i2s_last_timing(const i2s_ctx* ctx, float ms[5]);

/* ---- multi-GPU: one process per GPU, contiguous shards of the batch, no data-path collective; the only
 * exchange is ONE all-gather of the i2s_board records, device-resident on both sides, over RCCL (xGMI inside a node).
 *   rank 0:      i2s_comm_unique_id(id), then hand the 128 bytes to every rank by any host channel (torch.distributed
 *                broadcast, a file, MPI ...);
 *   every rank:  i2s_comm_create(&comm, device, id, world, rank, records_per_rank)  -- collective (ncclCommInitRank);
 *                the communicator owns a device buffer [world][records_per_rank] of records; i2s_comm_shard() is this
 *                rank's part of it, i2s_comm_all() the whole;
 *                i2s_set_board_sink(ctx, shard + first)  -- detect calls of ctx then also leave image i's record at
 *                sink[i] on the device (NULL switches it off), so a batch detected by several contexts / calls lands in
 *                the shard without touching the host;
 *                i2s_allgather_boards(ctx, comm, d_boards, n_local, d_all, h_all) -- ncclAllGather on ctx's stream of
 *                records_per_rank records per rank from d_boards (NULL = the shard: in place) into d_all (NULL = the
 *                communicator's buffer); d_boards need only hold the n_local <= records_per_rank records that are valid on
 *                this rank (records from elsewhere are copied into the shard first; the tail of the shard is zeroed); h_all,
 *                if not NULL, receives a host copy of all [world][records_per_rank] records.
 *                NOTE: a d_boards that is not the shard therefore OVERWRITES the shard -- including records a board sink
 *                (i2s_set_board_sink(shard + k)) has delivered there: use one way of filling the shard per gather, not both.
 *                The gather is the in-place form (sendbuff == recvbuff + rank * count) exactly when d_all is NULL or i2s_comm_all().
 *                Synchronous on return.  Rank r's records are those of images shard_range(total, r, world).
 * librccl is opened with dlopen on first use; I2S_E_NO_DEVICE if it is missing.  i2s_comm_create is collective even in failure:
 * a rank whose local allocation fails still takes part in ncclCommInitRank and reports afterwards.
 * i2s_comm_last_error(NULL) = text of this thread's last failed i2s_comm_unique_id / i2s_comm_create (dlopen, RCCL, HIP). */
#define I2S_COMM_ID_BYTES 128
typedef struct i2s_comm i2s_comm;
int  i2s_comm_unique_id(uint8_t id[I2S_COMM_ID_BYTES]);
int  i2s_comm_create(i2s_comm** out, int device_id, const uint8_t id[I2S_COMM_ID_BYTES], int world, int rank,
                     int records_per_rank);
void i2s_comm_destroy(i2s_comm* comm);
const char* i2s_comm_last_error(const i2s_comm* comm);
i2s_board* i2s_comm_shard(i2s_comm* comm);
i2s_board* i2s_comm_all(i2s_comm* comm);
int  i2s_set_board_sink(i2s_ctx* ctx, i2s_board* d_sink);
int  i2s_allgather_boards(i2s_ctx* ctx, i2s_comm* comm, const i2s_board* d_boards, int n_local,
                          i2s_board* d_all, i2s_board* h_all);

/* Per-kernel timing (bench.py's roofline objects): with profiling on, every kernel launch of a detect call is stamped with a pair of
 * HIP events on the context's stream -- the kernel's own start and stop (hipExtLaunchKernelGGL), i.e. the duration a rocprofv3
 * kernel trace reports, without the dispatch gaps between kernels; i2s_last_kernel_timing returns the milliseconds each of the
 * I2S_NSEG groups took, summed over the passes of the last detect call; i2s_kernel_timing_name(i) names group i. */
#define I2S_NSEG 14
int  i2s_set_profiling(i2s_ctx* ctx, int on);
int  i2s_last_kernel_timing(const i2s_ctx* ctx, float ms[I2S_NSEG]);
const char* i2s_kernel_timing_name(int i);
/* The blur bank's two-valued speculation (k_blur) on the last device pass: of the `total` 256 x 64-pixel bands of its images, `flagged`
 * held a pixel that is neither 0 nor 255 and went through the general kernels (Gaussians + sorting network + bit-sliced medians). */
int  i2s_blur_band_stats(i2s_ctx* ctx, int* flagged, int* total);
/* Canny hysteresis (cv.Canny's stack walk, img2sgf.py:162 and inside cv.HoughCircles :180) since i2s_create: `passes` device passes were
 * run, `redone` of them a second time because a phase's persistent tail gave up (pass budget or grid-barrier timeout) before the
 * fixed point; used_max[0 / 1] = the most propagation passes the main Canny's map / HoughCircles' maps have needed in one device pass. */
int  i2s_hysteresis_stats(const i2s_ctx* ctx, long long* passes, long long* redone, int used_max[2]);

/* Debug/test hooks (not part of the drop-in surface): Hough-circle accumulator of variant v
 * ((h)x(w) int32, cell layout = pixel layout) and line accumulators. Enabled by
 * i2s_set_debug(ctx, 1) before the detect call. */
int  i2s_set_debug(i2s_ctx* ctx, int on);
int  i2s_fetch_circle_acc(i2s_ctx* ctx, int index, int variant, int32_t* dst);
int  i2s_fetch_line_acc(i2s_ctx* ctx, int index, int32_t* dst, size_t cap, int* numrho, int* nangles);

#ifdef __cplusplus
}
#endif
#endif /* I2S_H */
