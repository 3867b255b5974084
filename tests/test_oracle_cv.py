"""Known-answer tests for the CPU restatement of the OpenCV rows (SURVEY Appendix A KATs) and the
reference's single recorded end-to-end result (ex1.jpg, screenshot-settings.jpg / screenshot.jpg)."""
import os

import numpy as np

from oracle import cv_oracle as cvo
from oracle import pipeline
from helpers import GOLDEN

EX1_SGF = "(;GM[1]FF[4]SZ[19]\nPL[W]\nAW[cn][jq][nq][qf][qj]\nAB[co][dd][dp][fp][nd][pd][pn][pp][ql]\n)\n"


def test_grey_kat():
    px = np.array([[[255, 0, 0], [0, 0, 255], [0, 255, 0], [17, 17, 17]]], np.uint8)
    assert cvo.bgr2gray(px).tolist() == [[29, 76, 150, 17]]
    assert cvo.bgr2gray(px, 14).tolist()[0][3] == 17


def test_gauss_kernels():
    assert cvo.gauss_kernel_q8(3, 3).tolist() == [84, 88, 84]
    assert cvo.gauss_kernel_q8(5, 5).tolist() == [49, 52, 54, 52, 49]
    assert cvo.gauss_kernel_q8(7, 7).tolist() == [35, 36, 38, 38, 38, 36, 35]


def test_gauss_kat():
    img = np.full((9, 9), 77, np.uint8)
    for k in (1, 3, 5, 7):
        assert (cvo.gaussian_blur(img, k, k) == 77).all()
    img = np.zeros((9, 9), np.uint8)
    img[4, 4] = 255
    out = cvo.gaussian_blur(img, 3, 3)
    assert out[4, 4] == 30 and out[4, 3] == 29 and out[3, 3] == 27 and out[4, 6] == 0
    # REFLECT_101 at the border: impulse at column 1 is seen twice by column 0's window? no: -1 -> 1
    img = np.zeros((5, 5), np.uint8)
    img[2, 1] = 255
    out = cvo.gaussian_blur(img, 3, 3)
    assert out[2, 0] == (255 * (84 + 84) * 88 + 32768) >> 16


def test_median_kat():
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (23, 31), dtype=np.uint8)
    for k in (3, 5, 7):
        r = k // 2
        pad = np.pad(img, r, mode="edge")
        exp = np.empty_like(img)
        for y in range(img.shape[0]):
            for x in range(img.shape[1]):
                exp[y, x] = np.sort(pad[y:y + k, x:x + k].ravel())[k * k // 2]
        np.testing.assert_array_equal(cvo.median_blur(img, k), exp)
    np.testing.assert_array_equal(cvo.median_blur(img, 1), img)


def test_canny_kat():
    assert (cvo.canny(np.full((16, 16), 99, np.uint8)) == 0).all()
    img = np.zeros((16, 16), np.uint8)
    img[:, 8:] = 255                       # vertical step between columns 7 and 8
    e = cvo.canny(img, 50, 200)
    # dx = 1020 at columns 7 and 8; the '>' / '>=' asymmetry keeps exactly the left one
    assert (e[:, 7] == 255).all() and (e[:, 8] == 0).all() and e.sum() == 255 * 16
    # 3-channel: the channel with the largest magnitude decides
    rgb = np.zeros((16, 16, 3), np.uint8)
    rgb[:, 8:, 2] = 255
    np.testing.assert_array_equal(cvo.canny(rgb, 50, 200), e)


def test_hysteresis_weak_chain():
    # a ramp edge that is strong only in its upper half: weak part survives through connectivity
    img = np.zeros((40, 40), np.uint8)
    img[:20, 20:] = 255                    # strong: mag 1020
    img[20:, 20:] = 30                     # weak: mag 120 (> 50, <= 200)
    e, m = cvo.canny(img, 50, 200, return_map=True)
    assert (m == 0).any()
    assert e[35, 19] == 255                # weak pixel far from the strong part, linked through the chain
    img2 = np.zeros((40, 40), np.uint8)
    img2[20:, 20:] = 30                    # same weak edge without any strong seed
    assert (cvo.canny(img2, 50, 200) == 0).all()


def test_hough_circles_kat():
    img = np.full((201, 201), 255, np.uint8)
    yy, xx = np.mgrid[0:201, 0:201]
    img[(xx - 100) ** 2 + (yy - 100) ** 2 <= 400] = 0
    # a binary disc's Sobel directions are too coarse for param2=30 (centre votes smear over ~7x7 cells,
    # max 19): nothing is found, exactly the behaviour the reference's blur bank exists to work around
    assert len(cvo.hough_circles(img)) == 0
    c = cvo.hough_circles(cvo.gaussian_blur(img, 5, 5))
    assert len(c) == 1
    assert 99.5 <= c[0, 0] <= 100.5 and 99.5 <= c[0, 1] <= 100.5 and abs(c[0, 2] - 20) <= 1
    assert len(cvo.hough_circles(np.full((64, 64), 10, np.uint8))) == 0


def test_erase_kat():
    img = np.full((40, 40), 255, np.uint8)
    out = cvo.erase_circles(img, np.array([[20.5, 20.5, 5.0]], np.float32))
    # r+2 = 7: box corners round(13.5)=14 .. round(27.5)=28 (half-even), plus at (20,20)
    assert (out[14:29, 14:29] == 0).sum() == 15 * 15 - 5
    assert out[20, 20] == 255 and out[19, 20] == 255 and out[20, 21] == 255 and out[19, 19] == 0
    assert out[13, 20] == 255 and out[29, 20] == 255
    # a later rectangle erases an earlier dot
    out = cvo.erase_circles(img, np.array([[20.5, 20.5, 5.0], [22.5, 20.5, 5.0]], np.float32))
    assert out[20, 20] == 0 and out[20, 22] == 255


def test_hough_lines_kat():
    import math
    img = np.zeros((100, 120), np.uint8)
    img[37, 10:90] = 255
    d = math.pi / 180
    for mode in (0, 1):
        l = cvo.hough_lines(img, 1, d, 40, math.pi / 2 - d, math.pi / 2 + d, mode)
        assert l is not None and 37.0 in l[:, 0, 0].tolist()
    assert cvo.hough_lines(img, 1, d, 100, math.pi / 2 - d, math.pi / 2 + d) is None
    assert cvo.hough_numangle(math.pi / 2 - d, math.pi / 2 + d, d, 0) == 3
    assert cvo.hough_numangle(math.pi / 2 - d, math.pi / 2 + d, d, 1) == 2
    assert cvo.hough_numangle(0, d, d, 0) == 2
    assert cvo.hough_numangle(0, d, d, 1) == 1


def test_ex1_known_answer():
    """The reference's only recorded result: ex1.jpg at defaults -> Hough threshold 74, 9 black + 5 white,
    white to play (screenshot-settings.jpg, screenshot.jpg; SGF layout from img2sgf.py:781-810)."""
    img = pipeline.load_and_enhance(os.path.join(GOLDEN, "test_images", "ex1.jpg"))
    r = pipeline.process_image(img, keep_planes=False)
    assert r["threshold"] == 74
    assert (r["num_black_stones"], r["num_white_stones"]) == (9, 5)
    assert r["sgf"] == EX1_SGF


# ---- independent implementations (scipy is in the image; cv2 is not) --------------------------------------------------------

def _awkward_images():
    rng = np.random.default_rng(5)
    yield rng.integers(0, 256, (37, 53), dtype=np.uint8)
    yield np.where(rng.random((40, 41)) < 0.5, 0, 255).astype(np.uint8)              # two-valued (the diagrams)
    yield (rng.integers(0, 4, (29, 64)) * 85).astype(np.uint8)                        # few levels
    yield rng.integers(0, 256, (1, 9), dtype=np.uint8)
    yield rng.integers(0, 256, (9, 1), dtype=np.uint8)
    yield rng.integers(0, 256, (2, 2), dtype=np.uint8)


def test_median_against_scipy():
    """cv.medianBlur (img2sgf.py:174) = exact median with BORDER_REPLICATE = scipy.ndimage.median_filter(mode='nearest')."""
    from scipy import ndimage
    for img in _awkward_images():
        for k in (3, 5, 7):
            np.testing.assert_array_equal(cvo.median_blur(img, k), ndimage.median_filter(img, size=k, mode="nearest"),
                                          err_msg="k=%d shape %s" % (k, img.shape))


def test_sobel_against_scipy():
    """The Sobel 3x3 of Canny / HoughCircles (CV_16S, BORDER_REPLICATE) = correlation with the Sobel kernels, mode 'nearest'."""
    from scipy import ndimage
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.int32)
    for img in _awkward_images():
        dx, dy = cvo.sobel3(img)
        i32 = img.astype(np.int32)
        np.testing.assert_array_equal(dx, ndimage.correlate(i32, kx, mode="nearest"))
        np.testing.assert_array_equal(dy, ndimage.correlate(i32, kx.T, mode="nearest"))


def test_gaussian_against_scipy():
    """cv.GaussianBlur 8-bit fixed-point path (img2sgf.py:175): integer taps, BORDER_REFLECT_101 (= scipy 'mirror'),
    (sum + 32768) >> 16 -- recomputed with scipy's separable correlation in int64."""
    from scipy import ndimage
    for img in _awkward_images():
        for k in (3, 5, 7):
            taps = cvo.gauss_kernel_q8(k, k).astype(np.int64)
            t = ndimage.correlate1d(img.astype(np.int64), taps, axis=1, mode="mirror")
            a = ndimage.correlate1d(t, taps, axis=0, mode="mirror")
            np.testing.assert_array_equal(cvo.gaussian_blur(img, k, k), ((a + 32768) >> 16).astype(np.uint8),
                                          err_msg="k=%d shape %s" % (k, img.shape))


def test_grey_against_numpy():
    rng = np.random.default_rng(6)
    rgb = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8).astype(np.int64)
    want = (rgb[..., 0] * 3735 + rgb[..., 1] * 19235 + rgb[..., 2] * 9798 + 16384) >> 15
    np.testing.assert_array_equal(cvo.bgr2gray(rgb.astype(np.uint8)), want.astype(np.uint8))


def test_hough_lines_row_and_column_counts():
    """At theta = pi/2 (sin = 1, |cos| < 1e-7) rho is the row index and at theta = 0 the column index: the accumulators of those
    angles are plain row / column counts of the non-zero pixels."""
    import math
    rng = np.random.default_rng(8)
    img = (rng.random((50, 70)) < 0.2).astype(np.uint8) * 255
    d = math.pi / 180
    _, dbg = cvo.hough_lines(img, 1, d, 1, 0.0, d, 0, debug=True)
    numrho = 2 * (50 + 70) + 1
    acc = np.asarray(dbg["acc"]).reshape(-1, numrho + 2)
    cols = (img != 0).sum(axis=0)
    np.testing.assert_array_equal(acc[1, 1 + (numrho - 1) // 2: 1 + (numrho - 1) // 2 + 70], cols)


# ---- committed digests of the oracle's answers (tests/golden/oracle_stage_digests.json) ---------------------------------------

def test_oracle_matches_its_committed_digests():
    """The digest file any cv2 owner can check (python -m oracle.cv2_harness --digests) is what the oracle answers today."""
    import sys
    sys.path.insert(0, GOLDEN)
    import make_oracle_digests as mk
    from oracle import stage_digests as sd
    doc = sd.load()
    assert len(doc["inputs"]) == 21
    fresh = mk.build()
    assert fresh["inputs"] == doc["inputs"]
    assert doc["inputs"]["ex1.jpg"]["sgf"] == EX1_SGF
