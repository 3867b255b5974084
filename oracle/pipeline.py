"""Headless CPU run of the reference's board-detection path  --  TEST INFRASTRUCTURE ONLY.

process_image() below follows img2sgf.py:117-204 + find_grid 546-576 step by step using
the CPU restatement of the OpenCV calls (cv_oracle) and of the glue (glue).  It is the
checker for the HIP path and the "port" CPU baseline of bench.py; the product never
imports it.
"""
import numpy as np

from . import cv_oracle as cvo
from . import glue

MAXBLUR = 3  # img2sgf.py:51


def load_and_enhance(path, contrast=70, brightness=50):
    """img2sgf.py:651 (open, convert RGB), 136-150 (contrast/brightness via Pillow).
    rotate_angle = 0 and the full-image selection (616-640 defaults) are identities."""
    from PIL import Image, ImageEnhance
    im = Image.open(path).convert("RGB")
    im = im.rotate(angle=-0, fillcolor="white",
                   center=((0 + im.size[0]) / 2, 0 + im.size[1] / 2)).crop((0, 0) + im.size)
    im = ImageEnhance.Contrast(im).enhance(102 / (101 - contrast) - 1)
    im = ImageEnhance.Brightness(im).enhance(450 / (200 - brightness) - 2)
    return np.array(im)


def blur_bank(grey, edges, compat):
    """img2sgf.py:171-175: [grey, edges, median1, gauss1, median3, gauss3, ...]."""
    blurs = [grey, edges]
    for i in range(MAXBLUR + 1):
        b = 2 * i + 1
        blurs.append(cvo.median_blur(grey, b))
        blurs.append(cvo.gaussian_blur(grey, b, b, compat["gauss_kernel_mode"]))
    return blurs


def process_image(img, threshold=None, black_thr=128, alignment=(glue.LEFT, glue.TOP),
                  compat=None, keep_planes=True, canny=(50, 200), hc=(10, 100, 30, 1, 30)):
    """img: HxW (grey) or HxWx3 (RGB as the reference holds it) uint8, i.e. the array
    `input_image_np` of img2sgf.py:150.  Returns a dict with every value the reference
    leaves in its globals after process_image()/find_grid()."""
    compat = dict(cvo.DEFAULT_COMPAT, **(compat or {}))
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape[:2]
    if threshold is None:
        threshold = glue.choose_threshold(W, H)
    out = dict(threshold=threshold)
    grey = cvo.bgr2gray(img, compat["grey_shift"])                     # :153
    edges = cvo.canny(img, canny[0], canny[1])                         # :162 (50, 200 in the reference)
    blurs = blur_bank(grey, edges, compat)                             # :171-175
    per_variant = []
    circles = np.zeros((0, 3), np.float32)
    for b in blurs:                                                    # :179-186
        c = cvo.hough_circles(b, *hc)                                  # (10, 100, 30, 1, 30) in the reference
        per_variant.append(c)
        if len(c) > 0:
            circles = np.vstack((circles, c))
    removed = cvo.erase_circles(edges, circles)                        # :188-198
    out.update(circles_all=circles, circles_per_variant=per_variant)
    if keep_planes:
        out.update(grey=grey, edges=edges, blurs=blurs, circles_removed=removed)
    # find_grid (:546-576)
    nm = compat["houghlines_numangle"]
    hlines = glue.find_lines(removed, threshold, True, nm)             # :259
    vlines = glue.find_lines(removed, threshold, False, nm)            # :261
    hcentres = glue.cluster_centres(hlines)                            # :298-299
    vcentres = glue.cluster_centres(vlines)                            # :301-302
    found_grid = len(hcentres) > 0 and len(vcentres) > 0               # :329
    g = glue.validate_grid(hcentres, vcentres, list(circles))          # :554
    out.update(hlines=np.asarray(hlines, np.float32).reshape(-1),
               vlines=np.asarray(vlines, np.float32).reshape(-1),
               hcentres=hcentres, vcentres=vcentres, found_grid=found_grid,
               valid_grid=g["valid"], hsize=g["hsize"], vsize=g["vsize"],
               hcentres_complete=g["hc"], vcentres_complete=g["vc"],
               hspace=g["hspace"], vspace=g["vspace"],
               circles=np.array(g["circles"], np.float32).reshape(-1, 3),
               board_ready=False, full_board=None, sgf=None)
    if g["valid"] and g["hsize"] <= glue.BOARD_SIZE and g["vsize"] <= glue.BOARD_SIZE:   # :568-574
        ib = glue.identify_board(grey, g, black_thr, alignment)
        out.update(ib)
        out["board_ready"] = True
        out["sgf"] = glue.to_sgf(ib["full_board"], ib["side_to_move"])
    return out
