"""Host-side mirror of the reference's board-detection interface (img2sgf.py:103-596) on top of the HIP
kernels.  Same function names as the reference, explicit arguments and return values instead of module
globals; `Detection` carries exactly the globals the reference's GUI code reads afterwards.

    det = process_image(input_image_np)            # img2sgf.py:117-204 + find_grid 546-576
    det.circles, det.hcentres_complete, det.full_board, to_SGF(det.full_board, det.side_to_move) ...

Everything numeric runs on the GPU through the C ABI (include/i2s.h); there is no CPU fallback.
"""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import I2sBoard, I2sError, I2sParams, I2sResult, I2sXform, PLANE_NAMES, STATUS_TEXT

BOARD_SIZE = 19                       # img2sgf.py:43
EMPTY, BLACK, WHITE, STONE = range(4)  # BoardStates, img2sgf.py:82-83
TOP, BOTTOM, LEFT, RIGHT = range(4)    # Alignment, img2sgf.py:86-87


@dataclass
class Params:
    """The reference's constants (img2sgf.py:43-57) and hard-wired OpenCV call arguments."""
    canny_lo: int = 50                 # edge_min_default :47
    canny_hi: int = 200                # edge_max_default :48
    hc_min_dist: float = 10.0          # HoughCircles args :180
    hc_param1: int = 100
    hc_param2: int = 30
    hc_min_radius: int = 1
    hc_max_radius: int = 30
    line_threshold: int = 0            # 0 -> choose_threshold(image) (:638)
    black_threshold: int = 128         # black_stone_threshold_default :45
    alignment: Sequence[int] = (LEFT, TOP)   # board_alignment :627
    min_grid_spacing: float = 10       # :54
    big_space_ratio: float = 1.6       # :55
    angle_tolerance: float = 1.0       # :52 (degrees)
    # OpenCV-version switches.  The reference pins no OpenCV version; the defaults restate OpenCV 4.3 .. 4.5.1 (2020, the
    # year the reference was written: DESIGN.md 2a has the evidence); Params.for_opencv(cv2.__version__) gives the set of any release.
    grey_shift: int = 15               # 15: OpenCV 4.x coefficients | 14: 3.x
    gauss_kernel_mode: int = 0         # 0: 8-bit taps diffused to sum 256 (>= 4.3 / 3.4.10) | 1: each tap rounded on its own (4.0 - 4.2)
    houghlines_numangle_mode: int = 1  # 1: cvRound(range / theta) angles (<= 4.5.1) | 0: floor(range / theta) + 1 (>= 4.5.2)
    contrast: Optional[int] = None     # 0..100: ImageEnhance.Contrast on the device (img2sgf.py:141-144); None = input is already enhanced
    brightness: Optional[int] = None   # 0..100: ImageEnhance.Brightness on the device (:146-149)
    schedule: bool = False             # ragged batches larger than one device pass: form the passes over images sorted by area
    jpeg_entropy_device: int = 1       # detect_jpeg, Huffman decoding: 0 host threads, 1 sequential files on the device, 2 all on the device

    @staticmethod
    def opencv_switches(version: str) -> dict:
        """The three version switches for an OpenCV release string such as cv2.__version__ ("4.2.0", "4.8.1.78", "3.4.9").
        Best knowledge of OpenCV's history, unpinned like the rest of rows a2-a8 (no cv2 in the build image):
        BGR2GRAY went from 14-bit to 15-bit coefficients with 4.0; GaussianBlur's 8-bit taps are error-diffused to sum 256 since
        4.3.0 / 3.4.10; HoughLines counts its angles with floor(..) + 1 since 4.5.2 / 3.4.14.
        THE RELEASE BOUNDARIES (4.3.0 / 3.4.10, 4.5.2 / 3.4.14) ARE FROM MEMORY AND UNVERIFIED: no changelog and no cv2 were within
        reach of the build.  Wherever a live cv2 module is at hand, use Params.from_cv2(cv2) / probe_cv2_switches(cv2) instead: they
        decide by what the module computes, not by what it is called."""
        v = tuple(int("".join(ch for ch in part if ch.isdigit()) or 0) for part in (version.split(".") + ["0", "0"])[:3])
        three = v[0] < 4
        return dict(grey_shift=14 if three else 15,
                    gauss_kernel_mode=0 if (v >= (4, 3, 0) or (three and v >= (3, 4, 10))) else 1,
                    houghlines_numangle_mode=0 if (v >= (4, 5, 2) or (three and v >= (3, 4, 14))) else 1)

    @classmethod
    def for_opencv(cls, version: str, **kwargs):
        """Params whose version switches restate the given OpenCV release (see opencv_switches)."""
        return cls(**dict(cls.opencv_switches(version), **kwargs))

    @classmethod
    def from_cv2(cls, cv, **kwargs):
        """Params whose version switches restate what the given, live cv2 module COMPUTES (probe_cv2_switches): the robust form of
        for_opencv(cv2.__version__).  Raises I2sError if the module answers a probe in a way none of the known rules explains."""
        return cls(**dict(probe_cv2_switches(cv), **kwargs))

    def switch_set(self) -> dict:
        return dict(grey_shift=self.grey_shift, gauss_kernel_mode=self.gauss_kernel_mode,
                    houghlines_numangle_mode=self.houghlines_numangle_mode)

    def to_c(self, inputs_on_device=False):
        p = I2sParams()
        p.canny_lo, p.canny_hi = self.canny_lo, self.canny_hi
        p.hc_min_dist = self.hc_min_dist
        p.hc_param1, p.hc_param2 = self.hc_param1, self.hc_param2
        p.hc_min_radius, p.hc_max_radius = self.hc_min_radius, self.hc_max_radius
        p.line_threshold, p.black_threshold = self.line_threshold, self.black_threshold
        p.align_x, p.align_y = int(self.alignment[0]), int(self.alignment[1])
        p.min_grid_spacing, p.big_space_ratio = self.min_grid_spacing, self.big_space_ratio
        p.angle_tolerance_deg = self.angle_tolerance
        p.grey_shift, p.gauss_kernel_mode = self.grey_shift, self.gauss_kernel_mode
        p.houghlines_numangle_mode = self.houghlines_numangle_mode
        p.inputs_on_device = 1 if inputs_on_device else 0
        p.contrast = -1 if self.contrast is None else int(self.contrast)
        p.brightness = -1 if self.brightness is None else int(self.brightness)
        p.schedule = 1 if self.schedule else 0
        p.jpeg_entropy_device = int(self.jpeg_entropy_device)
        return p


def _gauss_taps_8bit(k: int, mode: int):
    """The 8.8 fixed-point taps of cv.GaussianBlur(.., (k, k), k) (getGaussianKernelBitExact + getGaussianKernelFixedPoint_ED), as
    csrc/i2s_api.hip gauss_taps computes them: mode 0 error-diffused to sum 256, mode 1 each tap rounded on its own."""
    import math
    sig = float(k)
    scale2x = -0.125 / (sig * sig)
    vals = [math.exp(float(x * x) * scale2x) for x in range(1 - k, 0, 2)]
    mul = 1.0 / (sum(vals) * 2 + 1.0)
    kern = [v * mul for v in vals]
    if mode == 1:
        half = [int(round(v * 256.0)) for v in kern]
        return half + [int(round(mul * 256.0))] + half[::-1]
    half, err = [], 0.0
    for v in kern:
        adj = v * 256.0 + err
        v0 = int(round(adj))                   # Python's round and lrint both round half to even
        err = adj - v0
        half.append(v0)
    return half + [256 - 2 * sum(half)] + half[::-1]


def probe_cv2_switches(cv) -> dict:
    """The three OpenCV-version switches read off a LIVE cv2 module by three closed-form probes on the very calls the
    reference makes (img2sgf.py:153, 175, 236-244) -- what the module computes, not what its version string says:

    * cv.cvtColor(rgb, COLOR_BGR2GRAY) on 20 fixed colours: (3735 c0 + 19235 c1 + 9798 c2 + 2^14) >> 15 (4.x) or
      (1868 c0 + 9617 c1 + 4899 c2 + 2^13) >> 14 (3.x) -- the two differ on 16 of them (and on only 0.5 % of all colours);
    * cv.GaussianBlur((k, k), k), k = 3, 5, 7, of (a) a 255 impulse: out[dy][dx] = (t_dy t_dx 255 + 2^15) >> 16, and (b) one row of 128
      in a black image: every pixel of row c + d = (t_d 128 S + 2^15) >> 16 with S the tap sum -- (b) tells the two tap sets apart at
      every k (k = 3: centre 44 | 45; k = 5: 27 | 26; k = 7, d = 2: 18 | 19), (a) confirms the two-pass 8.8 fixed-point arithmetic;
    * cv.HoughLines of an image with TWO adjacent pixels at threshold 1 with the reference's three (min_theta, max_theta) pairs: at every
      probed angle the pair shares one rho bin and that bin is a local maximum (the pixels sit 200 px from both axes, so neighbouring
      angles hit different rho bins), hence the number of lines returned IS numangle: (2, 1, 1) = cvRound(range / theta), (3, 2, 2) =
      floor(range / theta) + 1.

    An answer none of the rules explains raises I2sError -- never a guess.  Self-contained: the checker package of the tests is not involved."""
    import math
    # the two coefficient sets agree on 99.5 % of all colours (2 x 1868 = 3735 + 1, 2 x 9617 = 19235 - 1): 16 colours on which they do
    # not, and white / black / two greys on which they must
    px = np.array([[0, 47, 90], [11, 90, 85], [23, 139, 215], [36, 207, 110], [51, 169, 15], [67, 231, 235], [86, 211, 160], [106, 196, 135],
                   [127, 95, 245], [148, 204, 120], [168, 231, 250], [187, 239, 110], [204, 72, 20], [219, 41, 35], [232, 99, 60], [244, 123, 15],
                   [255, 255, 255], [0, 0, 0], [128, 128, 128], [77, 77, 77]], np.uint8).reshape(4, 5, 3)
    c = px.astype(np.int64)
    g15 = ((3735 * c[..., 0] + 19235 * c[..., 1] + 9798 * c[..., 2] + (1 << 14)) >> 15).astype(np.uint8)
    g14 = ((1868 * c[..., 0] + 9617 * c[..., 1] + 4899 * c[..., 2] + (1 << 13)) >> 14).astype(np.uint8)
    assert (g15 != g14).sum() == 16
    got = np.asarray(cv.cvtColor(px, cv.COLOR_BGR2GRAY))
    if got.shape == g15.shape and np.array_equal(got, g15):
        grey_shift = 15
    elif got.shape == g14.shape and np.array_equal(got, g14):
        grey_shift = 14
    else:
        raise I2sError("cv2 probe: cvtColor(BGR2GRAY) follows neither the 15-bit nor the 14-bit coefficients")

    n, cc = 17, 8
    imp = np.zeros((n, n), np.uint8)
    imp[cc, cc] = 255
    row = np.zeros((n, n), np.uint8)
    row[cc, :] = 128
    fits = []
    for mode in (0, 1):
        ok = True
        for k in (3, 5, 7):
            t = np.array(_gauss_taps_8bit(k, mode), np.int64)
            r = k // 2
            want_imp = np.zeros((n, n), np.int64)
            want_imp[cc - r:cc + r + 1, cc - r:cc + r + 1] = (np.outer(t, t) * 255 + 32768) >> 16
            want_row = np.zeros((n, n), np.int64)
            want_row[cc - r:cc + r + 1, :] = ((t * 128 * int(t.sum()) + 32768) >> 16)[:, None]
            ok &= np.array_equal(np.asarray(cv.GaussianBlur(imp, (k, k), k)), want_imp)
            ok &= np.array_equal(np.asarray(cv.GaussianBlur(row, (k, k), k)), want_row)
        fits.append(ok)
    if fits[0] == fits[1]:
        raise I2sError("cv2 probe: GaussianBlur's impulse / line responses match neither 8-bit tap set (sum-256 error-diffused, "
                       "plainly rounded)")
    gauss_kernel_mode = 0 if fits[0] else 1

    # two adjacent pixels at (200, 200): side by side for the near-horizontal call, one above the other for the two near-vertical ones --
    # at every probed angle the pair falls into ONE rho bin (2 votes, above threshold 1; a threshold of 0 might be refused) and
    # neighbouring angles hit different bins (rho moves by 3.5 per degree 200 px from the axes), so each angle yields exactly one line
    delta = math.pi / 180.0 * Params.angle_tolerance                     # angle_delta, img2sgf.py:52
    counts = []
    for (lo, hi), second in (((math.pi / 2 - delta, math.pi / 2 + delta), (200, 201)), ((0.0, delta), (201, 200)),
                             ((math.pi - delta, math.pi), (201, 200))):
        pair = np.zeros((256, 256), np.uint8)
        pair[200, 200] = 255
        pair[second] = 255
        lines = cv.HoughLines(pair, rho=1, theta=math.pi / 180.0, threshold=1, min_theta=lo, max_theta=hi)
        counts.append(0 if lines is None else len(lines))
    if counts == [2, 1, 1]:
        numangle_mode = 1
    elif counts == [3, 2, 2]:
        numangle_mode = 0
    else:
        raise I2sError("cv2 probe: HoughLines answers the reference's three angle ranges with %s angles; known: [2, 1, 1] and "
                       "[3, 2, 2]" % counts)
    return dict(grey_shift=grey_shift, gauss_kernel_mode=gauss_kernel_mode, houghlines_numangle_mode=numangle_mode)


@dataclass
class Detection:
    """What the reference leaves in its globals after process_image()."""
    status: int
    status_text: str
    threshold: int
    circles_all: np.ndarray            # `circles` after the HoughCircles loop (:186), float32 (n,3)
    n_per_slot: List[int]
    circles: np.ndarray                # after validate_grid's radius filter (:443)
    hlines: np.ndarray                 # raw rho lists from find_all_lines (:258-265)
    vlines: np.ndarray
    hcentres: np.ndarray
    vcentres: np.ndarray
    found_grid: bool
    valid_grid: bool
    board_ready: bool
    hsize: int
    vsize: int
    hcentres_complete: Optional[np.ndarray]
    vcentres_complete: Optional[np.ndarray]
    hspace: Optional[float]
    vspace: Optional[float]
    detected_board: Optional[np.ndarray]   # (hsize, vsize)
    full_board: Optional[np.ndarray]       # (19, 19), [column, row]
    stone_brightnesses: np.ndarray
    num_black_stones: int
    num_white_stones: int
    side_to_move: int                  # 1 black, 2 white (img2sgf.py:89)
    planes: dict = field(default_factory=dict)   # optional: grey / edges / removed numpy images

    @property
    def sgf(self):
        return to_SGF(self.full_board, self.side_to_move) if self.board_ready else None


def _detection_from_result(r: I2sResult) -> Detection:
    n = r.n_circles
    circ = np.ctypeslib.as_array(r.circles)[:n].copy()
    kept = np.ctypeslib.as_array(r.circle_kept)[:n].astype(bool)
    valid = bool(r.valid_grid)
    ready = bool(r.board_ready)
    hs, vs = r.hsize, r.vsize
    full = np.ctypeslib.as_array(r.board).astype(np.float64).copy() if ready else None
    det = np.ctypeslib.as_array(r.detected)[:hs, :vs].astype(np.float64).copy() if ready else None
    return Detection(
        status=r.status, status_text=STATUS_TEXT.get(r.status, "?"), threshold=r.line_threshold,
        circles_all=circ, n_per_slot=list(r.n_per_slot), circles=circ[kept] if valid else circ,
        hlines=np.ctypeslib.as_array(r.hlines)[:r.n_hlines].copy(),
        vlines=np.ctypeslib.as_array(r.vlines)[:r.n_vlines].copy(),
        hcentres=np.ctypeslib.as_array(r.hcentres)[:r.n_hcentres].copy(),
        vcentres=np.ctypeslib.as_array(r.vcentres)[:r.n_vcentres].copy(),
        found_grid=bool(r.found_grid), valid_grid=valid, board_ready=ready, hsize=hs, vsize=vs,
        hcentres_complete=np.ctypeslib.as_array(r.hcentres_complete)[:r.n_hcomplete].copy() if valid else None,
        vcentres_complete=np.ctypeslib.as_array(r.vcentres_complete)[:r.n_vcomplete].copy() if valid else None,
        hspace=r.hspace if valid else None, vspace=r.vspace if valid else None,
        detected_board=det, full_board=full,
        stone_brightnesses=np.ctypeslib.as_array(r.brightness)[:r.n_stones].copy() if ready else np.zeros(0),
        num_black_stones=r.n_black, num_white_stones=r.n_white, side_to_move=r.side_to_move)


class _LazyShapes:
    """(h, w) of the files of the last pass, parsed on first use (fetch_source / fetch_plane need them, a bare detect does not)."""

    def __init__(self, det, blobs):
        self._det, self._blobs, self._shapes = det, list(blobs), None

    def __getitem__(self, i):
        if self._shapes is None:
            self._shapes = [self._det.jpeg_info(b)[1::-1] for b in self._blobs]
        return self._shapes[i]

    def __len__(self):
        return len(self._blobs)


def jpeg_info(data: bytes, lib=None):
    """(w, h, components) of a JPEG that i2s_detect_jpeg_batch can decode (8-bit, Huffman-coded, sequential or progressive, grey or YCbCr);
    raises I2sError for anything else (arithmetic, CMYK, not a JPEG).  Host-only: no GPU context needed."""
    lib = lib if lib is not None else _lib.load()
    w, h, c = C.c_int(), C.c_int(), C.c_int()
    rc = lib.dll.i2s_jpeg_info(data, len(data), C.byref(w), C.byref(h), C.byref(c))
    if rc != 0:
        raise I2sError(lib.dll.i2s_strerror(rc).decode())
    return w.value, h.value, c.value


class Detector:
    """One GPU context (one HIP stream on one device).  max_batch = images per device pass."""

    def __init__(self, device=0, max_batch=16, max_w=1024, max_h=1024, lib=None):
        self.lib = lib if lib is not None else _lib.load()
        self._ctx = C.c_void_p()
        rc = self.lib.dll.i2s_create(C.byref(self._ctx), device, max_batch, max_w, max_h)
        if rc != 0:
            raise I2sError("i2s_create failed: %s (no CPU fallback exists)" % self.lib.dll.i2s_strerror(rc).decode())
        self.max_batch, self.max_w, self.max_h = max_batch, max_w, max_h
        self._last_shapes = []
        self.arch = _lib.device_arch(self.lib, self._ctx)       # "gfx950..." on an MI355X ("emulated": tests/emu)

    def close(self):
        if self._ctx:
            self.lib.dll.i2s_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise I2sError("%s: %s" % (self.lib.dll.i2s_strerror(rc).decode(),
                                       self.lib.dll.i2s_last_error(self._ctx).decode()))

    # -- raw entry: pointers may be host (numpy) or device addresses
    def detect_ptrs(self, ptrs, ws, hs, strides, chans, params: Params, on_device, full=False, xforms=None):
        """xforms: None, or one (affine[6], crop[4]) per image (preprocess.xform): rotate + crop on the device first."""
        B = len(ptrs)
        arr_p = (C.c_void_p * B)(*[int(p) for p in ptrs])
        mk = lambda v: (C.c_int * B)(*[int(x) for x in v])
        boards = (I2sBoard * B)()
        res = (I2sResult * B)() if full else None
        p = params.to_c(on_device)
        shapes = list(zip(hs, ws))
        if xforms is None:
            rc = self.lib.dll.i2s_detect_batch(self._ctx, B, arr_p, mk(ws), mk(hs), mk(strides), mk(chans), C.byref(p),
                                               boards, res)
        else:
            xf = (I2sXform * B)()
            for i, (aff, crop) in enumerate(xforms):
                xf[i].affine[:] = [float(v) for v in aff]
                xf[i].crop[:] = [int(v) for v in crop]
            shapes = [(x.crop[3] - x.crop[1], x.crop[2] - x.crop[0]) for x in xf]          # the cropped regions
            rc = self.lib.dll.i2s_detect_batch_xf(self._ctx, B, arr_p, mk(ws), mk(hs), mk(strides), mk(chans), xf,
                                                  C.byref(p), boards, res)
        self._check(rc)
        n_last = (B - 1) % self.max_batch + 1 if B else 0
        if params.schedule and B > self.max_batch:
            # the C side formed its passes over the images sorted by area (stable): the last pass holds the largest ones
            order = sorted(range(B), key=lambda i: shapes[i][0] * shapes[i][1])
            self._last_shapes = [shapes[i] for i in order[B - n_last:]]
        else:
            self._last_shapes = shapes[B - n_last:]
        return boards, res

    def detect_batch(self, images: Sequence[np.ndarray], params: Optional[Params] = None, full=True, xforms=None):
        """images: HxW (grey) or HxWx3 uint8 arrays = the reference's `input_image_np` (img2sgf.py:150) -- or, with
        `xforms` (one preprocess.xform(...) per image), the decoded source images, which are then rotated and cropped on the
        device first (crop_and_rotate_image, img2sgf.py:110-114).
        Returns a list of Detection (full=True) or the raw I2sBoard array (full=False)."""
        params = params or Params()
        imgs = [np.ascontiguousarray(im, np.uint8) for im in images]
        for im in imgs:
            if im.ndim not in (2, 3) or (im.ndim == 3 and im.shape[2] != 3):
                raise ValueError("images must be HxW or HxWx3 uint8")
        boards, res = self.detect_ptrs([im.ctypes.data for im in imgs], [im.shape[1] for im in imgs],
                                       [im.shape[0] for im in imgs], [im.strides[0] for im in imgs],
                                       [1 if im.ndim == 2 else 3 for im in imgs], params, False, full, xforms)
        if not full:
            return boards
        return [_detection_from_result(r) for r in res]

    def jpeg_info(self, data: bytes):
        """(w, h, components) of a JPEG the device path can decode; raises I2sError (unsupported / invalid) otherwise."""
        return jpeg_info(data, self.lib)

    def jpeg_last_rounds(self) -> int:
        """Rounds the parallel entropy decoder's iteration took in the last pass (0 = it did not run)."""
        return int(self.lib.dll.i2s_jpeg_last_rounds(self._ctx))

    def jpeg_last_handed_back(self) -> int:
        """Files of the last JPEG pass the parallel entropy decoder handed to the serial one at its round limit."""
        return int(self.lib.dll.i2s_jpeg_last_handed_back(self._ctx))

    def jpeg_set_max_rounds(self, rounds: int):
        """Limit of the parallel entropy decoder's iteration; a pass that needs more goes to the serial decoder."""
        self._check(self.lib.dll.i2s_jpeg_set_max_rounds(self._ctx, int(rounds)))

    def jpeg_last_timing(self):
        """Host wall times of the last detect_jpeg call, ms: parsing, entropy stage host work, entropy stage device wait, whole call."""
        ms = (C.c_float * 4)()
        self._check(self.lib.dll.i2s_jpeg_last_timing(self._ctx, ms))
        return list(ms)

    def detect_jpeg(self, blobs: Sequence[bytes], params: Optional[Params] = None, full=True, xforms=None):
        """blobs: the bytes of JPEG files (8-bit, Huffman-coded, sequential or progressive).  Image.open(path).convert("RGB") (img2sgf.py:651) happens on the device
        (Huffman stage per Params.jpeg_entropy_device), bit-exact with Pillow; then as detect_batch (xforms / Params.contrast / .brightness apply
        to the decoded image).  Raises I2sError("parameter outside the supported envelope") for CMYK, arithmetic-coded
        and other flavours -- decode those with Pillow and call detect_batch."""
        params = params or Params()
        B = len(blobs)
        arr = (C.c_char_p * B)(*blobs)
        lens = (C.c_size_t * B)(*[len(b) for b in blobs])
        boards = (I2sBoard * B)()
        res = (I2sResult * B)() if full else None
        xf = None
        if xforms is not None:
            xf = (I2sXform * B)()
            for i, (aff, crop) in enumerate(xforms):
                xf[i].affine[:] = [float(v) for v in aff]
                xf[i].crop[:] = [int(v) for v in crop]
            shapes = [(x.crop[3] - x.crop[1], x.crop[2] - x.crop[0]) for x in xf]
        else:
            shapes = None                       # frame sizes: read from the files only where (and when) they are needed
        p = params.to_c()
        self._check(self.lib.dll.i2s_detect_jpeg_batch(self._ctx, B, arr, lens, xf, C.byref(p), boards, res))
        n_last = (B - 1) % self.max_batch + 1 if B else 0
        if params.schedule and B > self.max_batch:
            if shapes is None:
                shapes = [self.jpeg_info(b)[1::-1] for b in blobs]
            order = sorted(range(B), key=lambda i: shapes[i][0] * shapes[i][1])
            self._last_shapes = [shapes[i] for i in order[B - n_last:]]
        elif shapes is None:
            self._last_shapes = _LazyShapes(self, blobs[B - n_last:])
        else:
            self._last_shapes = shapes[B - n_last:]
        if not full:
            return boards
        return [_detection_from_result(r) for r in res]

    def set_board_sink(self, device_ptr):
        """Detect calls then also leave image i's 384-byte record at device_ptr + 384 i ON THE DEVICE (None / 0 = off):
        how a rank's shard of the multi-GPU gather buffer is filled without touching the host (dist.BoardGather.sink)."""
        self._check(self.lib.dll.i2s_set_board_sink(self._ctx, C.c_void_p(int(device_ptr) if device_ptr else None)))

    def detect_device(self, batch, params: Optional[Params] = None, sink=None):
        """batch: a torch uint8 tensor (B,H,W) or (B,H,W,3) resident on this context's GPU.  Returns (B,) I2sBoard
        array; the pixels are read in place (no copy).  sink: see set_board_sink (for this call only)."""
        if sink is not None:
            self.set_board_sink(sink)
            try:
                return self.detect_device(batch, params)
            finally:
                self.set_board_sink(None)
        params = params or Params()
        assert batch.is_cuda and batch.is_contiguous() and str(batch.dtype) == "torch.uint8"
        B, H, W = batch.shape[:3]
        cn = 1 if batch.dim() == 3 else 3
        # single-channel sources with dword-aligned rows ARE their grey planes (no copy): classify() and fetch_plane("grey")
        # read these pixels after the call has returned, so the tensor must outlive it -- until the next detect call
        self._last_batch = batch
        base, step = batch.data_ptr(), H * W * cn
        boards, _ = self.detect_ptrs([base + i * step for i in range(B)], [W] * B, [H] * B, [W * cn] * B, [cn] * B,
                                     params, True, False)
        return boards

    def classify(self, first, n, params: Params, full=True):
        """identify_board() only (apply_black_thresh, img2sgf.py:762-766) on images of the last device pass."""
        boards = (I2sBoard * n)()
        res = (I2sResult * n)() if full else None
        p = params.to_c()
        self._check(self.lib.dll.i2s_classify_batch(self._ctx, first, n, C.byref(p), boards, res))
        return [_detection_from_result(r) for r in res] if full else boards

    def validate_grid_raw(self, hcentres, vcentres, circles, params: Optional[Params] = None):
        """i2s_validate_grid: validate_grid() (img2sgf.py:420-445) on float64 centres; returns the I2sResult record."""
        params = params or Params()
        hc = np.ascontiguousarray(hcentres if hcentres is not None else [], np.float64).reshape(-1)
        vc = np.ascontiguousarray(vcentres if vcentres is not None else [], np.float64).reshape(-1)
        c = np.ascontiguousarray(circles, np.float32).reshape(-1, 3)
        f64p, f32p = C.POINTER(C.c_double), C.POINTER(C.c_float)
        res = I2sResult()
        p = params.to_c()
        self._check(self.lib.dll.i2s_validate_grid(self._ctx, hc.ctypes.data_as(f64p), len(hc), vc.ctypes.data_as(f64p), len(vc),
                                                    c.ctypes.data_as(f32p), len(c), C.byref(p), C.byref(res)))
        return res

    def grid_from_lines(self, grey, circles, hlines, vlines, params: Optional[Params] = None):
        """find_grid() (img2sgf.py:546-576) on injected circles and rho lists."""
        params = params or Params()
        grey = np.ascontiguousarray(grey, np.uint8)
        c = np.ascontiguousarray(circles, np.float32).reshape(-1, 3)
        hl = np.ascontiguousarray(hlines, np.float32).reshape(-1)
        vl = np.ascontiguousarray(vlines, np.float32).reshape(-1)
        f32p, u8p = C.POINTER(C.c_float), C.POINTER(C.c_uint8)
        board, res = I2sBoard(), I2sResult()
        p = params.to_c()
        self._check(self.lib.dll.i2s_grid_from_lines(
            self._ctx, grey.ctypes.data_as(u8p), grey.shape[1], grey.shape[0], c.ctypes.data_as(f32p), len(c),
            hl.ctypes.data_as(f32p), len(hl), vl.ctypes.data_as(f32p), len(vl), C.byref(p), C.byref(board),
            C.byref(res)))
        self._last_shapes = [grey.shape]
        return _detection_from_result(res)

    def find_all_lines(self, circles_removed, params: Optional[Params] = None):
        """find_all_lines() (img2sgf.py:258-265): (hlines, vlines) float32 rho lists of an injected circles_removed image."""
        params = params or Params()
        img = np.ascontiguousarray(circles_removed, np.uint8)
        if img.ndim != 2:
            raise ValueError("circles_removed must be HxW uint8")
        hl = np.zeros(_lib.MAX_LINES, np.float32)
        vl = np.zeros(_lib.MAX_LINES, np.float32)
        nh, nv = C.c_int(0), C.c_int(0)
        f32p = C.POINTER(C.c_float)
        p = params.to_c()
        self._check(self.lib.dll.i2s_find_lines(self._ctx, img.ctypes.data_as(C.POINTER(C.c_uint8)), img.shape[1], img.shape[0],
                                                img.strides[0], C.byref(p), hl.ctypes.data_as(f32p), C.byref(nh),
                                                vl.ctypes.data_as(f32p), C.byref(nv)))
        self._last_shapes = [img.shape]
        return hl[:nh.value].copy(), vl[:nv.value].copy()

    def fetch_plane(self, index, plane):
        """numpy image of a device plane of the last pass: 'grey', 'edges', 'removed', 'median3', ... or an int id."""
        pid = PLANE_NAMES[plane] if isinstance(plane, str) else int(plane)
        h, w = self._last_shapes[index][:2]
        out = np.empty((h, w), np.uint8)
        self._check(self.lib.dll.i2s_fetch_plane(self._ctx, index, pid, out.ctypes.data_as(C.POINTER(C.c_uint8)), w))
        return out

    def fetch_source(self, index, channels=3):
        """`input_image_np` (img2sgf.py:150) of image `index` of the last pass, after the on-device enhancement."""
        h, w = self._last_shapes[index][:2]
        out = np.empty((h, w, channels) if channels > 1 else (h, w), np.uint8)
        self._check(self.lib.dll.i2s_fetch_source(self._ctx, index, out.ctypes.data_as(C.POINTER(C.c_uint8)), w * channels))
        return out

    def set_debug(self, on=True):
        self._check(self.lib.dll.i2s_set_debug(self._ctx, 1 if on else 0))

    def fetch_circle_acc(self, index, variant):
        h, w = self._last_shapes[index][:2]
        out = np.empty((h, w), np.int32)
        self._check(self.lib.dll.i2s_fetch_circle_acc(self._ctx, index, variant, out.ctypes.data_as(C.POINTER(C.c_int32))))
        return out

    def fetch_line_acc(self, index):
        h, w = self._last_shapes[index][:2]
        nr = 2 * (w + h) + 1
        out = np.zeros((12, nr), np.int32)
        numrho, nang = C.c_int(0), (C.c_int * 3)()
        self._check(self.lib.dll.i2s_fetch_line_acc(self._ctx, index, out.ctypes.data_as(C.POINTER(C.c_int32)),
                                                    out.size, C.byref(numrho), nang))
        return out, list(nang)

    def last_timing(self):
        ms = (C.c_float * 5)()
        self._check(self.lib.dll.i2s_last_timing(self._ctx, ms))
        return dict(blur_canny_ms=ms[0], hough_circles_ms=ms[1], erase_lines_ms=ms[2], grid_ms=ms[3], total_ms=ms[4])


    def set_profiling(self, on=True):
        """Per-kernel HIP events on this context's stream (include/i2s.h: i2s_set_profiling)."""
        self._check(self.lib.dll.i2s_set_profiling(self._ctx, 1 if on else 0))

    def blur_band_stats(self):
        """(flagged, total): bands of the last device pass that were NOT two-valued (went through the general blur / median kernels)."""
        f, t = C.c_int(), C.c_int()
        self._check(self.lib.dll.i2s_blur_band_stats(self._ctx, C.byref(f), C.byref(t)))
        return f.value, t.value

    def hysteresis_stats(self):
        """dict(passes, redone, used_max): device passes since the context was created, how many of them had to be run again because
        Canny's hysteresis had not converged, and the most propagation passes (main Canny, HoughCircles) a device pass has needed."""
        n, r, k = C.c_longlong(), C.c_longlong(), (C.c_int * 2)()
        self._check(self.lib.dll.i2s_hysteresis_stats(self._ctx, C.byref(n), C.byref(r), k))
        return dict(passes=n.value, redone=r.value, used_max=[k[0], k[1]])

    def last_kernel_timing(self):
        """{kernel group: ms} of the last detect call, in launch order (profiling must be on)."""
        ms = (C.c_float * _lib.NSEG)()
        self._check(self.lib.dll.i2s_last_kernel_timing(self._ctx, ms))
        return {self.lib.dll.i2s_kernel_timing_name(i).decode(): float(ms[i]) for i in range(_lib.NSEG)}


class StreamedDetector:
    """n_streams independent contexts (one HIP stream + workspace each) on one GPU, driven from n_streams host threads
    (ctypes releases the GIL during the C call).  A device-resident batch is split into contiguous slices, one per
    stream, so the latency-bound tail kernels of one slice (grid repair, sorting, peak search) overlap with the
    throughput-bound kernels of another."""

    def __init__(self, device=0, n_streams=2, max_batch=64, max_w=1024, max_h=1024, lib=None):
        from concurrent.futures import ThreadPoolExecutor
        self.dets = [Detector(device, max_batch, max_w, max_h, lib=lib) for _ in range(n_streams)]
        self.pool = ThreadPoolExecutor(max_workers=n_streams)
        self.max_batch = max_batch

    def jpeg_info(self, data: bytes):
        """(w, h, components) of a JPEG the device path can decode; raises I2sError (unsupported / invalid) otherwise."""
        return jpeg_info(data, self.dets[0].lib)

    def detect_device(self, batch, params: Optional[Params] = None, sink=None):
        """sink: device address of the record of batch[0] (Detector.set_board_sink); every slice deposits at its offset."""
        n = len(self.dets)
        B = batch.shape[0]
        cuts = [B * i // n for i in range(n + 1)]
        futs = [self.pool.submit(self.dets[i].detect_device, batch[cuts[i]:cuts[i + 1]], params,
                                 None if sink is None else sink + cuts[i] * C.sizeof(I2sBoard))
                for i in range(n) if cuts[i + 1] > cuts[i]]
        out = (I2sBoard * B)()
        pos = 0
        for f in futs:
            part = f.result()
            C.memmove(C.byref(out, pos * C.sizeof(I2sBoard)), part, C.sizeof(part))
            pos += len(part)
        return out

    def detect_batch(self, images: Sequence[np.ndarray], params: Optional[Params] = None, xforms=None):
        """Host images (ragged sizes allowed): the passes are formed over the images sorted by area and dealt
        to the streams round-robin, so uploads of one stream overlap with kernels of another and every stream gets small
        and large passes alike.  Returns the (B,) I2sBoard array in input order."""
        B, n, mb = len(images), len(self.dets), self.max_batch
        def area(i):
            if xforms is not None:
                c = xforms[i][1]
                return (c[2] - c[0]) * (c[3] - c[1])
            return images[i].shape[0] * images[i].shape[1]
        order = sorted(range(B), key=area)
        passes = [order[k:k + mb] for k in range(0, B, mb)]
        share = [[i for ps in passes[s::n] for i in ps] for s in range(n)]
        def run(s):
            idx = share[s]
            xf = [xforms[i] for i in idx] if xforms is not None else None
            return self.dets[s].detect_batch([images[i] for i in idx], params, full=False, xforms=xf)
        futs = [(s, self.pool.submit(run, s)) for s in range(n) if share[s]]
        out = (I2sBoard * B)()
        for s, f in futs:
            part = f.result()
            for j, i in enumerate(share[s]):
                out[i] = part[j]
        return out

    def detect_jpeg(self, blobs: Sequence[bytes], params: Optional[Params] = None, xforms=None):
        """JPEG file bytes (Detector.detect_jpeg) over the streams: the passes are formed over the files sorted by pixel count
        and dealt round-robin, so the host-side Huffman decoding of one stream overlaps the kernels of another."""
        B, n, mb = len(blobs), len(self.dets), self.max_batch
        def area(i):
            if xforms is not None:
                c = xforms[i][1]
                return (c[2] - c[0]) * (c[3] - c[1])
            w, h, _ = jpeg_info(blobs[i], self.dets[0].lib)
            return w * h
        order = sorted(range(B), key=area)
        passes = [order[k:k + mb] for k in range(0, B, mb)]
        share = [[i for ps in passes[s::n] for i in ps] for s in range(n)]
        def run(s):
            idx = share[s]
            xf = [xforms[i] for i in idx] if xforms is not None else None
            return self.dets[s].detect_jpeg([blobs[i] for i in idx], params, full=False, xforms=xf)
        futs = [(s, self.pool.submit(run, s)) for s in range(n) if share[s]]
        out = (I2sBoard * B)()
        for s, f in futs:
            part = f.result()
            for j, i in enumerate(share[s]):
                out[i] = part[j]
        return out

    def last_timing(self):
        t = [d.last_timing() for d in self.dets]
        return {k: sum(x[k] for x in t) for k in t[0]}

    def close(self):
        for d in self.dets:
            d.close()
        self.pool.shutdown()


_default_detector = None


def _detector_for(images, detector):
    global _default_detector
    if detector is not None:
        return detector
    hmax = max(im.shape[0] for im in images)
    wmax = max(im.shape[1] for im in images)
    d = _default_detector
    if d is None or d.max_w < wmax or d.max_h < hmax:
        if d is not None:
            d.close()
        d = _default_detector = Detector(0, 8, max(wmax, 1024), max(hmax, 1024))
    return d


# ---- the reference's entry points (same names) ---------------------------------------------------------

def choose_threshold(img):
    """img2sgf.py:606-613.  img: PIL image, numpy image or (w, h)."""
    if hasattr(img, "size") and not isinstance(img, np.ndarray):
        w, h = img.size
    elif isinstance(img, np.ndarray):
        h, w = img.shape[:2]
    else:
        w, h = img
    return _lib.load().dll.i2s_choose_threshold(int(w), int(h))


_HINTED = False


def _hint_if_installed_cv2_differs(params):
    """the package defaults restate OpenCV 4.3 .. 4.5.1; a caller who has ALREADY imported a cv2 that computes something
    else (>= 4.5.2 counts one more angle per HoughLines call) and runs on the defaults is told once, and how to follow the module.
    cv2 is never imported here, and nothing is changed: a hint, not a decision."""
    global _HINTED
    import sys
    cv = sys.modules.get("cv2")
    if _HINTED or params is not None or cv is None:
        return
    _HINTED = True
    try:
        theirs = probe_cv2_switches(cv)
    except Exception:
        return
    if theirs != Params().switch_set():
        import warnings
        warnings.warn("img2sgf_amd runs on its default OpenCV switch set %s (OpenCV 4.3 .. 4.5.1), the cv2 imported in this process computes %s: "
                      "pass Params.from_cv2(cv2) to get the boards that cv2 would give (DESIGN.md 2a)" % (Params().switch_set(), theirs), stacklevel=3)


def process_image(input_image_np, params: Optional[Params] = None, detector: Optional[Detector] = None,
                  keep_planes=False) -> Detection:
    """img2sgf.py:117-204 from `input_image_np` (:150) on, including find_grid() (:546-576)."""
    _hint_if_installed_cv2_differs(params)
    d = _detector_for([input_image_np], detector)
    det = d.detect_batch([input_image_np], params, full=True)[0]
    if keep_planes:
        det.planes = {k: d.fetch_plane(0, k) for k in ("grey", "edges", "removed")}
    return det


def identify_board(detector: Detector, index=0, black_threshold=128, alignment=(LEFT, TOP)) -> Detection:
    """img2sgf.py:497-543 re-run on a cached detection (what apply_black_thresh does, :762-766)."""
    return detector.classify(index, 1, Params(black_threshold=black_threshold, alignment=alignment))[0]


def find_grid(grey, circles, hlines, vlines, params: Optional[Params] = None, detector: Optional[Detector] = None):
    """img2sgf.py:546-576 with find_lines' outputs injected."""
    d = _detector_for([grey], detector)
    return d.grid_from_lines(grey, circles, hlines, vlines, params)


# Direction, img2sgf.py:74-80
HORIZONTAL, VERTICAL = 1, 2


def find_all_lines(circles_removed_image_np, threshold, params: Optional[Params] = None, detector: Optional[Detector] = None):
    """img2sgf.py:258-265: (hlines, vlines) as (n,1) float32 rho columns ([] when a direction has no line), the three
    cv.HoughLines calls of find_lines on the device."""
    import dataclasses
    d = _detector_for([circles_removed_image_np], detector)
    p = dataclasses.replace(params or Params(), line_threshold=int(threshold))
    hl, vl = d.find_all_lines(circles_removed_image_np, p)
    col = lambda a: a.reshape(-1, 1) if len(a) else []
    return col(hl), col(vl)


def find_lines(circles_removed_image_np, threshold, direction, params: Optional[Params] = None,
               detector: Optional[Detector] = None):
    """img2sgf.py:230-255 for one Direction (HORIZONTAL / VERTICAL): the (n,1) rho column, or []."""
    hl, vl = find_all_lines(circles_removed_image_np, threshold, params, detector)
    return hl if direction in (HORIZONTAL, "H", "HORIZONTAL") else vl


_DUMMY_GREY = np.zeros((16, 16), np.uint8)


def cluster_lines(hlines, vlines, params: Optional[Params] = None, detector: Optional[Detector] = None):
    """img2sgf.py:295-332 without the plotting: (hcentres, vcentres, found_grid) -- single-linkage clusters of the rho
    lists at distance_threshold = min_grid_spacing (find_clusters_fixed_threshold :268-280), centre = float32 mean
    (get_cluster_centres :283-292); found_grid = both non-empty (:329-330).  Runs on the device (i2s_grid_from_lines)."""
    d = _detector_for([_DUMMY_GREY], detector)
    det = d.grid_from_lines(_DUMMY_GREY, np.zeros((0, 3), np.float32), hlines, vlines, params)
    return det.hcentres, det.vcentres, det.found_grid


def validate_grid(hcentres, vcentres, circles, params: Optional[Params] = None, detector: Optional[Detector] = None):
    """img2sgf.py:420-445 on explicit cluster centres: (valid, newcircles, vsize, hsize, hcentres_complete,
    vcentres_complete, hspace, vspace), or (False, circles, 0, 0, None, None, None, None) -- also for centres the reference
    would reject (closer than min_grid_spacing, a single line ...).  Takes what the reference's function takes: float64
    centres of any spacing.  Runs on the device (i2s_validate_grid)."""
    p = params or Params()
    c = np.ascontiguousarray(circles, np.float32).reshape(-1, 3)
    d = _detector_for([_DUMMY_GREY], detector)
    r = d.validate_grid_raw(hcentres, vcentres, c, p)
    if not r.valid_grid:
        return (False, c if len(c) else [], 0, 0, None, None, None, None)
    kept = np.frombuffer(r.circle_kept, np.uint8, count=r.n_circles).astype(bool)
    return (True, c[kept], r.vsize, r.hsize, np.array(r.hcentres_complete[:r.n_hcomplete], np.float64),
            np.array(r.vcentres_complete[:r.n_vcomplete], np.float64), r.hspace, r.vspace)


def to_SGF(board, side_to_move):
    """img2sgf.py:781-810, byte for byte."""
    letters = "abcdefghijklmnopqrstuvwxyz"
    board = np.asarray(board)
    out = "(;GM[1]FF[4]SZ[" + str(BOARD_SIZE) + "]\n"
    out += "PL[B]\n" if side_to_move == 1 else "PL[W]\n"
    black_moves, white_moves = "", ""
    if (board == BLACK).any():
        black_moves = "AB" + "".join("[" + letters[i] + letters[j] + "]" for i in range(BOARD_SIZE)
                                     for j in range(BOARD_SIZE) if board[i, j] == BLACK)
    if (board == WHITE).any():
        white_moves = "AW" + "".join("[" + letters[i] + letters[j] + "]" for i in range(BOARD_SIZE)
                                     for j in range(BOARD_SIZE) if board[i, j] == WHITE)
    if side_to_move == 1:
        return out + black_moves + "\n" + white_moves + "\n" + ")\n"
    return out + white_moves + "\n" + black_moves + "\n" + ")\n"


def board_to_sgf(b: I2sBoard):
    """SGF text of a compact board record (None when no board was produced)."""
    if b.status != 0:
        return None
    return to_SGF(np.ctypeslib.as_array(b.board), b.side_to_move)
