"""Restatement of the reference's pure-Python/numpy glue  --  TEST INFRASTRUCTURE ONLY.

Stateless versions (explicit arguments instead of module globals) of
img2sgf.py:230-255 (find_lines), 268-292 (clustering), 335-445 (grid repair /
validation), 448-543 (snapping + stone classifier), 606-613 (choose_threshold) and
781-810 (to_SGF).  PINNED: tests/golden/glue_*.json hold input/output vectors produced
by importing the reference itself (tests/golden/make_glue_golden.py) and
tests/test_oracle_glue.py checks this file against them.
"""
import math
from bisect import bisect_left

import numpy as np

from . import cv_oracle as cvo

BOARD_SIZE = 19                  # img2sgf.py:43
MIN_GRID_SPACING = 10            # :54
BIG_SPACE_RATIO = 1.6            # :55
ANGLE_DELTA = math.pi / 180 * 1.0  # :52-53
EMPTY, BLACK, WHITE, STONE = range(4)   # :82-83
TOP, BOTTOM, LEFT, RIGHT = range(4)     # :86-87


def choose_threshold(w, h):
    """img2sgf.py:606-613."""
    t = int(min(w, h) / 12.8 + 16)
    return int(min(max(t, 20), 200))


def find_lines(circles_removed, threshold, horizontal, numangle_mode=None):
    """img2sgf.py:230-255 -> (n,1) float32 rho column, or [] when nothing is found.  numangle_mode None = the default switch set."""
    if numangle_mode is None:
        numangle_mode = cvo.DEFAULT_COMPAT["houghlines_numangle"]
    theta = math.pi / 180.0
    if horizontal:
        lines = cvo.hough_lines(circles_removed, 1, theta, threshold,
                                math.pi / 2 - ANGLE_DELTA, math.pi / 2 + ANGLE_DELTA, numangle_mode)
    else:
        v1 = cvo.hough_lines(circles_removed, 1, theta, threshold, 0, ANGLE_DELTA, numangle_mode)
        v2 = cvo.hough_lines(circles_removed, 1, theta, threshold, math.pi - ANGLE_DELTA, math.pi,
                             numangle_mode)
        if v2 is not None:
            v2[:, 0, 0] = -v2[:, 0, 0]
            v2[:, 0, 1] = v2[:, 0, 1] - math.pi
            lines = np.vstack((v1, v2)) if v1 is not None else v2
        else:
            lines = v1
    return [] if lines is None else lines[:, 0, 0].reshape(-1, 1)


def cluster_centres(rhos):
    """img2sgf.py:268-292: sklearn single linkage with distance_threshold=10 on a 1-D
    column == sort, split where the consecutive gap is >= 10; centre = float32 mean of
    the members, stored as float64, sorted.  < 2 samples -> sklearn raises -> []."""
    r = np.asarray(rhos, np.float32).reshape(-1)
    if r.size < 2:
        return np.zeros(0)
    s = np.sort(r)
    out = []
    start = 0
    for i in range(1, s.size + 1):
        if i == s.size or float(s[i]) - float(s[i - 1]) >= MIN_GRID_SPACING:
            out.append(np.float64(s[start:i].mean()))   # float32 mean
            start = i
    out = np.array(out, np.float64)
    out.sort()
    return out


def truncate_grid(x):
    """img2sgf.py:400-417."""
    if x is None:
        return None
    if len(x) == BOARD_SIZE + 2:
        return x[1:-1]
    if len(x) == BOARD_SIZE + 1:
        return x[:-1]
    return x


def complete_grid(x):
    """img2sgf.py:335-397."""
    if x is None or len(x) == 0 or len(x) == 1:
        return None
    spaces = x[1:] - x[:-1]
    min_space = min(spaces)
    if min_space < MIN_GRID_SPACING:
        return None
    bound = min_space * BIG_SPACE_RATIO
    big = spaces[spaces > bound]
    if len(big) == 0:
        return x
    small = spaces[spaces <= bound]
    max_space = max(small)
    avg = (min_space + max_space) / 2
    n = len(small)
    for s in big:
        n += int(round(s / avg))
    if n > BOARD_SIZE + 2:
        return None
    n += 1
    if len(x) < n:
        ans = np.zeros(n)
        ans[0] = x[0]
        i, j = 1, 1
        for s in spaces:
            if s <= max_space:
                ans[i] = x[j]
                i += 1
                j += 1
            else:
                m = int(round(s / avg))
                for k in range(m):
                    ans[i] = x[j - 1] + (k + 1) * s / m
                    i += 1
                j += 1
        return ans
    return x


def validate_grid(hcentres, vcentres, circles):
    """img2sgf.py:420-445.  Returns dict(valid, circles, vsize, hsize, hc, vc, hspace, vspace)."""
    bad = dict(valid=False, circles=circles, vsize=0, hsize=0, hc=None, vc=None, hspace=None, vspace=None)
    hc = truncate_grid(complete_grid(truncate_grid(hcentres)))
    if hc is None:
        return bad
    vc = truncate_grid(complete_grid(truncate_grid(vcentres)))
    if vc is None:
        return bad
    vsize, hsize = len(hc), len(vc)
    hspace = (hc[-1] - hc[0]) / vsize
    vspace = (vc[-1] - vc[0]) / hsize
    lo = min(hspace, vspace) * 0.3
    hi = max(hspace, vspace) * 0.65
    kept = [c for c in circles if lo < c[2] < hi]
    return dict(valid=True, circles=kept, vsize=vsize, hsize=hsize, hc=hc, vc=vc,
                hspace=hspace, vspace=vspace)


def closest_index(a, x):
    """img2sgf.py:448-459."""
    i = bisect_left(x, a)
    if i == 0:
        return 0
    if i == len(x):
        return i - 1
    return i - 1 if a - x[i - 1] <= x[i] - a else i


def window(i, j, g):
    """Window of average_intensity (img2sgf.py:468-480) -> (xmin, xmax, ymin, ymax), clipped."""
    x = g["vc"][i]
    xmin, xmax = int(round(x - g["hspace"] / 2)), int(round(x + g["hspace"] / 2))
    y = g["hc"][j]
    ymin, ymax = int(round(y - g["vspace"] / 2)), int(round(y + g["vspace"] / 2))
    H, W = g["shape"]
    return max(0, xmin), min(W, xmax), max(0, ymin), min(H, ymax)


def average_intensity(grey, i, j, g):
    """img2sgf.py:468-481."""
    xmin, xmax, ymin, ymax = window(i, j, g)
    sl = grey[ymin:ymax, xmin:xmax]
    if sl.size == 0:
        return float("nan")
    return float(np.mean(sl))


def identify_board(grey, g, black_thr=128, alignment=(LEFT, TOP)):
    """img2sgf.py:497-543 (+ align_board 484-494)."""
    hsize, vsize = g["hsize"], g["vsize"]
    g = dict(g, shape=grey.shape)
    det = np.zeros((hsize, vsize))
    for c in g["circles"]:
        det[closest_index(c[0], g["vc"]), closest_index(c[1], g["hc"])] = STONE
    br = []
    for j in range(hsize):
        for k in range(vsize):
            if det[j, k] == STONE:
                br.append(average_intensity(grey, j, k, g))
    br = np.array(br, np.float64)
    nblack = int(np.sum(br <= black_thr))
    nwhite = len(br) - nblack
    side = BLACK if nblack <= nwhite else WHITE      # 1 / 2 (img2sgf.py:89, 529-534)
    for i in range(hsize):
        for j in range(vsize):
            if det[i, j] == STONE:
                det[i, j] = BLACK if average_intensity(grey, i, j, g) <= black_thr else WHITE
    full = np.zeros((BOARD_SIZE, BOARD_SIZE))
    xo = BOARD_SIZE - hsize if alignment[0] == RIGHT else 0
    yo = BOARD_SIZE - vsize if alignment[1] == BOTTOM else 0
    full[xo:xo + hsize, yo:yo + vsize] = det
    return dict(detected_board=det, full_board=full, stone_brightnesses=br,
                num_black_stones=nblack, num_white_stones=nwhite, side_to_move=side)


def to_sgf(board, side_to_move):
    """img2sgf.py:781-810."""
    letters = "abcdefghijklmnopqrstuvwxyz"
    out = "(;GM[1]FF[4]SZ[%d]\n" % BOARD_SIZE
    out += "PL[B]\n" if side_to_move == 1 else "PL[W]\n"
    b, w = "", ""
    if (board == BLACK).any():
        b = "AB" + "".join("[%s%s]" % (letters[i], letters[j])
                           for i in range(BOARD_SIZE) for j in range(BOARD_SIZE) if board[i, j] == BLACK)
    if (board == WHITE).any():
        w = "AW" + "".join("[%s%s]" % (letters[i], letters[j])
                           for i in range(BOARD_SIZE) for j in range(BOARD_SIZE) if board[i, j] == WHITE)
    if side_to_move == 1:
        return out + b + "\n" + w + "\n)\n"
    return out + w + "\n" + b + "\n)\n"
