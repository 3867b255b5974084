"""ctypes front-end of oracle/libi2s_oracle.so  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Gives the CPU restatement of the OpenCV calls on the reference's hot path the same
call shape the reference uses (img2sgf.py:153-198, 236-244).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity status of these rows: "parity unpinned" (cv2 is absent; see i2s_oracle.c header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "libi2s_oracle.so")
    src = os.path.join(_HERE, "i2s_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libi2s_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        u8p, i16p, i32p, f32p, ip = (C.POINTER(C.c_uint8), C.POINTER(C.c_int16),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_float),
                                     C.POINTER(C.c_int))
        L.orc_bgr2gray.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.orc_canny.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u8p, u8p]
        L.orc_sobel3.argtypes = [u8p, C.c_int, C.c_int, C.c_int, i16p, i16p]
        L.orc_median.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.orc_gauss_kernel_q8.argtypes = [C.c_int, C.c_double, C.c_int, ip]
        L.orc_gauss.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, u8p, C.c_int]
        L.orc_hough_circles.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                        C.c_int, C.c_int, f32p, C.c_int, u8p, i32p, ip,
                                        f32p, C.c_int, ip]
        L.orc_hough_circles.restype = C.c_int
        L.orc_erase_circles.argtypes = [u8p, C.c_int, C.c_int, C.c_int, f32p, C.c_int]
        L.orc_hough_numangle.argtypes = [C.c_double, C.c_double, C.c_float, C.c_int]
        L.orc_hough_numangle.restype = C.c_int
        L.orc_hough_trig.argtypes = [C.c_int, C.c_double, C.c_float, C.c_float, f32p, f32p]
        L.orc_hough_lines.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int,
                                      C.c_double, C.c_double, C.c_int, f32p, C.c_int, i32p, ip]
        L.orc_hough_lines.restype = C.c_int
        L.orc_window_sum.argtypes = [u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_window_sum.restype = C.c_long
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


# Version switches; defaults = "current OpenCV 4.x".
# The defaults restate OpenCV 4.3 .. 4.5.1 (the reference is dated March 2020 and its own fixtures only give sane
# boards under the HoughLines angle count of those releases); the product's Params defaults are the same set.
DEFAULT_COMPAT = dict(grey_shift=15, gauss_kernel_mode=0, houghlines_numangle=1)
# The switch set tests/golden/oracle_stage_digests.json is written under (its "alt" entries hold the other value of each switch)
DIGEST_COMPAT = dict(grey_shift=15, gauss_kernel_mode=0, houghlines_numangle=0)


def bgr2gray(img, grey_shift=15):
    """cv.cvtColor(img, cv.COLOR_BGR2GRAY) for HxWx3 u8 (img2sgf.py:153)."""
    img = _u8(img)
    if img.ndim == 2:
        return img.copy()
    h, w, _ = img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_bgr2gray(_p(img, C.c_uint8), w, h, img.strides[0], _p(out, C.c_uint8), w, grey_shift)
    return out


def canny(img, low=50, high=200, return_map=False):
    """cv.Canny(img, low, high, apertureSize=3, L2gradient=False) (img2sgf.py:162)."""
    img = _u8(img)
    h, w = img.shape[:2]
    cn = 1 if img.ndim == 2 else img.shape[2]
    out = np.empty((h, w), np.uint8)
    m = np.empty((h, w), np.uint8)
    lib().orc_canny(_p(img, C.c_uint8), w, h, img.strides[0], cn, int(low), int(high),
                    _p(out, C.c_uint8), _p(m, C.c_uint8))
    return (out, m) if return_map else out


def sobel3(img):
    img = _u8(img)
    h, w = img.shape
    dx = np.empty((h, w), np.int16)
    dy = np.empty((h, w), np.int16)
    lib().orc_sobel3(_p(img, C.c_uint8), w, h, img.strides[0], _p(dx, C.c_int16), _p(dy, C.c_int16))
    return dx, dy


def median_blur(img, k):
    """cv.medianBlur(img, k) (img2sgf.py:174)."""
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_median(_p(img, C.c_uint8), w, h, img.strides[0], int(k), _p(out, C.c_uint8), w)
    return out


def gauss_kernel_q8(k, sigma, mode=0):
    taps = np.zeros(k, np.int32)
    rc = lib().orc_gauss_kernel_q8(int(k), float(sigma), int(mode), _p(taps, C.c_int))
    assert rc == 0
    return taps


def gaussian_blur(img, k, sigma, mode=0):
    """cv.GaussianBlur(img, (k,k), sigma) (img2sgf.py:175)."""
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    lib().orc_gauss(_p(img, C.c_uint8), w, h, img.strides[0], int(k), float(sigma), int(mode),
                    _p(out, C.c_uint8), w)
    return out


def hough_circles(img, min_dist=10, param1=100, param2=30, min_radius=1, max_radius=30, debug=False):
    """cv.HoughCircles(img, HOUGH_GRADIENT, 1, min_dist, [], param1, param2, minR, maxR)
    (img2sgf.py:180).  Returns (n,3) float32 (n may be 0)."""
    img = _u8(img)
    h, w = img.shape
    cap = 1 << 16
    out = np.zeros((cap, 3), np.float32)
    edges = np.zeros((h, w), np.uint8)
    acc = np.zeros((h + 2, w + 2), np.int32)
    est = np.zeros((cap, 4), np.float32)
    ncent = C.c_int(0)
    nest = C.c_int(0)
    n = lib().orc_hough_circles(_p(img, C.c_uint8), w, h, img.strides[0], float(min_dist),
                                int(round(param1)), int(round(param2)), int(min_radius), int(max_radius),
                                _p(out, C.c_float), cap, _p(edges, C.c_uint8), _p(acc, C.c_int32),
                                C.byref(ncent), _p(est, C.c_float), cap, C.byref(nest))
    assert n >= 0, "oracle circle capacity exceeded"
    res = out[:n].copy()
    if debug:
        return res, dict(edges=edges, acc=acc, n_centers=ncent.value, est=est[:nest.value].copy())
    return res


def erase_circles(img, circles):
    """img2sgf.py:191-198 (in place on a copy)."""
    out = _u8(img).copy()
    h, w = out.shape
    c = np.ascontiguousarray(circles, np.float32).reshape(-1, 3)
    if len(c):
        lib().orc_erase_circles(_p(out, C.c_uint8), w, h, out.strides[0], _p(c, C.c_float), len(c))
    return out


def hough_numangle(min_theta, max_theta, theta, mode=0):
    return lib().orc_hough_numangle(float(min_theta), float(max_theta), np.float32(theta), int(mode))


def hough_trig(numangle, min_theta, theta):
    s = np.zeros(max(numangle, 1), np.float32)
    c = np.zeros(max(numangle, 1), np.float32)
    lib().orc_hough_trig(int(numangle), float(min_theta), np.float32(theta), np.float32(1.0),
                         _p(s, C.c_float), _p(c, C.c_float))
    return s[:numangle], c[:numangle]


def hough_lines(img, rho, theta, threshold, min_theta, max_theta, numangle_mode=0, debug=False):
    """cv.HoughLines(img, rho, theta, threshold, min_theta=.., max_theta=..)
    (img2sgf.py:236-244).  Returns (n,1,2) float32 or None like cv2."""
    img = _u8(img)
    h, w = img.shape
    cap = 1 << 16
    out = np.zeros((cap, 2), np.float32)
    numangle = hough_numangle(min_theta, max_theta, theta, numangle_mode)
    numrho = 2 * (w + h) + 1
    acc = np.zeros((max(numangle, 0) + 2, numrho + 2), np.int32)
    na = C.c_int(0)
    n = lib().orc_hough_lines(_p(img, C.c_uint8), w, h, img.strides[0], np.float32(rho), np.float32(theta),
                              int(threshold), float(min_theta), float(max_theta), int(numangle_mode),
                              _p(out, C.c_float), cap, _p(acc, C.c_int32), C.byref(na))
    assert n >= 0
    res = out[:n].reshape(n, 1, 2).copy() if n > 0 else None
    if debug:
        return res, dict(acc=acc, numangle=na.value)
    return res
