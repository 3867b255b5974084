"""Plumbing check of the DORMANT cv2 harness (oracle/cv2_harness.py): cv2 is absent here, so a throw-away module object
that answers the ten calls from the oracle itself is put in sys.modules for the duration of one test.  This proves only
that the harness code runs end to end (argument shapes, None / empty handling, switch probing, the B1 timing loop) --
it says NOTHING about parity with OpenCV: the comparison is the oracle against itself.  The real check is
tests/test_cv2_crosscheck.py, which skips until a real cv2 is importable."""
import sys
import types

import numpy as np
import pytest

from img2sgf_amd import synth
from oracle import cv_oracle as cvo


def _oracle_backed_module(grey_shift=15, gauss_mode=0, numangle=0):
    m = types.ModuleType("cv2")
    m.__version__ = "0.0-oracle-backed-plumbing-stub"
    m.COLOR_BGR2GRAY, m.HOUGH_GRADIENT = 6, 3
    m.cvtColor = lambda img, code: cvo.bgr2gray(img, grey_shift)
    m.Canny = lambda img, lo, hi, apertureSize=3, L2gradient=False: cvo.canny(img, lo, hi)
    m.medianBlur = lambda g, k: cvo.median_blur(g, k)
    m.GaussianBlur = lambda g, ks, s: cvo.gaussian_blur(g, ks[0], s, gauss_mode)

    def hough_circles(b, method, dp, min_dist, circles, p1, p2, rmin, rmax):
        c = cvo.hough_circles(b, min_dist, p1, p2, rmin, rmax)
        return c.reshape(1, -1, 3) if len(c) else None
    m.HoughCircles = hough_circles

    def rectangle(img, ul, lr, colour, thickness):
        x0, y0, x1, y1 = max(ul[0], 0), max(ul[1], 0), min(lr[0], img.shape[1] - 1), min(lr[1], img.shape[0] - 1)
        if x0 <= x1 and y0 <= y1:
            img[y0:y1 + 1, x0:x1 + 1] = colour[0]

    def circle(img, c, r, colour, thickness):
        for dx, dy in ((0, 0), (1, 0), (-1, 0), (0, 1), (0, -1)):
            x, y = c[0] + dx, c[1] + dy
            if 0 <= x < img.shape[1] and 0 <= y < img.shape[0]:
                img[y, x] = colour[0]
    m.rectangle, m.circle = rectangle, circle
    m.HoughLines = lambda img, rho, theta, threshold, min_theta, max_theta: cvo.hough_lines(
        img, rho, theta, threshold, min_theta, max_theta, numangle)
    m.setNumThreads = lambda n: None
    m.getNumThreads = lambda: 1
    return m


@pytest.mark.parametrize("switches", [(15, 0, 0), (14, 1, 1)])
def test_harness_runs_and_selects_switches(monkeypatch, switches):
    monkeypatch.setitem(sys.modules, "cv2", _oracle_backed_module(*switches))
    from oracle import cv2_harness as H
    assert H.have_cv2()
    c = H.select_compat()
    assert (c["grey_shift"], c["gauss_kernel_mode"], c["houghlines_numangle"]) == switches
    img, _ = synth.synth_diagram(3, geom=synth.GEOM_SMALL)
    assert H.compare(img, c) == []
    rgb = np.repeat(img[:, :, None], 3, axis=2).copy()
    rgb[::7, ::5, 0] //= 2
    assert H.compare(rgb, c) == []
    r = H.cv2_process_image(img)
    assert r["board_ready"] and r["sgf"].startswith("(;GM[1]FF[4]SZ[19]")
    assert H.compare(np.zeros((40, 50), np.uint8), c) == []          # no circles, no lines: the None / empty paths


@pytest.mark.parametrize("switches,expect_bad", [((15, 0, 0), False), ((14, 1, 1), True)])
def test_digest_check_runs(monkeypatch, switches, expect_bad):
    """python -m oracle.cv2_harness --digests, with the oracle-backed stand-in as "cv2": under the default switches every digest
    matches; a stand-in that behaves like an older OpenCV is reported stage by stage, with the switch value it matches."""
    monkeypatch.setitem(sys.modules, "cv2", _oracle_backed_module(*switches))
    from oracle import cv2_harness as H
    lines = []
    bad = H.check_digests(out=lines.append)
    assert bool(bad) == expect_bad
    if expect_bad:
        assert any(b[2] == "gauss_kernel_mode=1" for b in bad) and any(b[2] == "grey_shift=14" for b in bad)
        assert "neither" not in lines[1]
