"""Live cross-check of the oracle against a real OpenCV  --  TEST INFRASTRUCTURE ONLY, dormant where cv2 is absent.

The reference's hot-path arithmetic lives in the third-party `cv2` module (opencv-python, version NOT pinned by the
reference; it only logs cv.__version__ at img2sgf.py:1246).  cv2 is not in this image, so rows a2-a8 of SURVEY 8 are
"parity unpinned".  The day a box has cv2, this module turns that into a pinned verdict:

  * `run_cv2_calls(img, threshold)` makes the ten calls exactly as the reference's call sites do --
    img2sgf.py:153 (cvtColor), :162-165 (Canny), :174 (medianBlur), :175 (GaussianBlur), :180 (HoughCircles),
    :197-198 (rectangle / circle), :236-244 (3x HoughLines) -- and returns every intermediate;
  * `select_compat()` probes the three version-sensitive choices of SURVEY Appendix A.7 (grey shift 15/14,
    Gaussian tap rounding, HoughLines numangle rule) against the installed cv2 and returns the switch set under
    which the oracle reproduces it (or raises, naming the first stage that no switch setting reproduces);
  * `compare(img, ...)` byte-compares oracle and cv2 stage by stage.

tests/test_cv2_crosscheck.py drives it over the 18 fixtures + synthetic diagrams; tools/cpu_baseline.py uses
`run_cv2_calls` for the B1 / B2 CPU baselines of BASELINE.md.  Only tests/, tools/cpu_baseline.py and bench.py's
cpu_baseline leg may import this module; the product never does.
"""
import math

import numpy as np

from . import cv_oracle as cvo
from . import glue

MAXBLUR = 3                                   # img2sgf.py:51
ANGLE_DELTA = math.pi / 180 * 1.0             # img2sgf.py:52-53


def have_cv2():
    try:
        import cv2  # noqa: F401
        return True
    except Exception:
        return False


def cv2_version():
    import cv2
    return cv2.__version__


def cv2_find_lines(cv, removed, threshold, horizontal):
    """The HoughLines calls of find_lines (img2sgf.py:230-255) -> (n,1) float32 rho column or []."""
    theta = math.pi / 180.0
    if horizontal:
        lines = cv.HoughLines(removed, rho=1, theta=theta, threshold=threshold,
                              min_theta=math.pi / 2 - ANGLE_DELTA, max_theta=math.pi / 2 + ANGLE_DELTA)
    else:
        v1 = cv.HoughLines(removed, rho=1, theta=theta, threshold=threshold, min_theta=0, max_theta=ANGLE_DELTA)
        v2 = cv.HoughLines(removed, rho=1, theta=theta, threshold=threshold, min_theta=math.pi - ANGLE_DELTA,
                           max_theta=math.pi)
        if v2 is not None:
            v2[:, 0, 0] = -v2[:, 0, 0]
            v2[:, 0, 1] = v2[:, 0, 1] - math.pi
            lines = np.vstack((v1, v2)) if v1 is not None else v2
        else:
            lines = v1
    return [] if lines is None else lines[:, 0, 0].reshape(-1, 1)


def run_cv2_calls(img, threshold=None, canny=(50, 200), with_lines=True):
    """img = `input_image_np` (img2sgf.py:150): HxW or HxWx3 uint8.  Returns a dict of every intermediate of
    img2sgf.py:153-198 and the rho lists of :258-265, computed by the installed cv2."""
    import cv2 as cv
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape[:2]
    if threshold is None:
        threshold = glue.choose_threshold(W, H)
    grey = cv.cvtColor(img, cv.COLOR_BGR2GRAY) if img.ndim == 3 else img.copy()      # :153 (C=1: benchmark input)
    edges = cv.Canny(img, canny[0], canny[1], apertureSize=3, L2gradient=False)     # :162-165
    blurs = [grey, edges]                                                            # :171-175
    for i in range(MAXBLUR + 1):
        b = 2 * i + 1
        blurs.append(cv.medianBlur(grey, b))
        blurs.append(cv.GaussianBlur(grey, (b, b), b))
    per_variant = []
    circles = np.zeros((0, 3), np.float32)
    for b in blurs:                                                                  # :179-186
        c = cv.HoughCircles(b, cv.HOUGH_GRADIENT, 1, 10, np.array([]), 100, 30, 1, 30)
        c = np.zeros((0, 3), np.float32) if c is None or len(c) == 0 else np.asarray(c[0], np.float32).reshape(-1, 3)
        per_variant.append(c)
        if len(c):
            circles = np.vstack((circles, c))
    removed = edges.copy()                                                           # :169, :188-198
    for xc, yc, r in circles:
        r = r + 2
        ul = (int(round(xc - r)), int(round(yc - r)))
        lr = (int(round(xc + r)), int(round(yc + r)))
        cv.rectangle(removed, ul, lr, (0, 0, 0), -1)
        cv.circle(removed, (int(round(xc)), int(round(yc))), 1, (255, 255, 255), -1)
    out = dict(threshold=threshold, grey=grey, edges=edges, blurs=blurs, circles_all=circles,
               circles_per_variant=per_variant, circles_removed=removed)
    if with_lines:
        out["hlines"] = np.asarray(cv2_find_lines(cv, removed, threshold, True), np.float32).reshape(-1)
        out["vlines"] = np.asarray(cv2_find_lines(cv, removed, threshold, False), np.float32).reshape(-1)
    return out


def cv2_process_image(img, threshold=None, black_thr=128, alignment=(glue.LEFT, glue.TOP)):
    """The reference's whole per-image path with the real cv2 for rows a2-a8 and the (pinned) glue for a9-a16,
    including find_clusters_fixed_threshold's SECOND round of HoughLines (img2sgf.py:269): what the reference costs on
    a CPU, and the SGF it would write."""
    import cv2 as cv
    r = run_cv2_calls(img, threshold)
    removed, thr = r["circles_removed"], r["threshold"]
    hl = cv2_find_lines(cv, removed, thr, True)            # the clustering functions call find_lines again (:269)
    vl = cv2_find_lines(cv, removed, thr, False)
    hc, vc = glue.cluster_centres(hl), glue.cluster_centres(vl)
    g = glue.validate_grid(hc, vc, list(r["circles_all"]))
    r.update(hcentres=hc, vcentres=vc, valid_grid=g["valid"], hsize=g["hsize"], vsize=g["vsize"], board_ready=False, sgf=None,
             full_board=None)
    if g["valid"] and g["hsize"] <= glue.BOARD_SIZE and g["vsize"] <= glue.BOARD_SIZE:
        ib = glue.identify_board(r["grey"], g, black_thr, alignment)
        r.update(ib)
        r["board_ready"] = True
        r["sgf"] = glue.to_sgf(ib["full_board"], ib["side_to_move"])
    return r


# ---- A.7 switch selection --------------------------------------------------------------------------------------------

def _probe_images():
    rng = np.random.default_rng(20240229)
    rgb = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    grey = rng.integers(0, 256, (67, 91), dtype=np.uint8)
    lines = np.zeros((120, 160), np.uint8)
    lines[30, 5:150] = 255
    lines[31, 20:140] = 255
    lines[5:110, 40] = 255
    lines[10:100, 41] = 255
    lines[8:115, 120] = 255
    return rgb, grey, lines


def select_compat():
    """Switch set (cv_oracle.DEFAULT_COMPAT keys) under which the oracle reproduces the installed cv2 on probe images.
    Raises AssertionError naming the stage if no setting does."""
    import cv2 as cv
    rgb, grey, lines = _probe_images()
    compat = {}
    want = cv.cvtColor(rgb, cv.COLOR_BGR2GRAY)
    for s in (15, 14):
        if np.array_equal(cvo.bgr2gray(rgb, s), want):
            compat["grey_shift"] = s
            break
    assert "grey_shift" in compat, "cvtColor BGR2GRAY: neither the 15-bit nor the 14-bit coefficients reproduce cv2 " + cv.__version__
    for m in (0, 1):
        if all(np.array_equal(cvo.gaussian_blur(grey, k, k, m), cv.GaussianBlur(grey, (k, k), k)) for k in (3, 5, 7)):
            compat["gauss_kernel_mode"] = m
            break
    assert "gauss_kernel_mode" in compat, "GaussianBlur: no tap-rounding mode reproduces cv2 " + cv.__version__
    theta = math.pi / 180.0
    ranges = ((math.pi / 2 - ANGLE_DELTA, math.pi / 2 + ANGLE_DELTA), (0.0, ANGLE_DELTA), (math.pi - ANGLE_DELTA, math.pi))
    for m in (0, 1):
        ok = True
        for lo, hi in ranges:
            a = cv.HoughLines(lines, rho=1, theta=theta, threshold=40, min_theta=lo, max_theta=hi)
            b = cvo.hough_lines(lines, 1, theta, 40, lo, hi, m)
            ok &= (a is None and b is None) or (a is not None and b is not None and np.array_equal(np.asarray(a, np.float32), b))
        if ok:
            compat["houghlines_numangle"] = m
            break
    assert "houghlines_numangle" in compat, "HoughLines: no numangle rule reproduces cv2 " + cv.__version__
    return compat


def compare(img, compat, threshold=None):
    """Stage-by-stage byte comparison of the oracle with cv2 on one image.  Returns a list of (stage, detail) for
    every stage that differs (empty = identical)."""
    from . import pipeline as opipe
    ref = run_cv2_calls(img, threshold)
    orc = opipe.process_image(img, threshold=threshold, compat=compat)
    bad = []

    def same(name, a, b):
        a, b = np.asarray(a), np.asarray(b)
        if a.shape != b.shape or not np.array_equal(a, b):
            n = int((a != b).sum()) if a.shape == b.shape else -1
            bad.append((name, "shape %s vs %s, %d differing" % (a.shape, b.shape, n)))

    same("cvtColor :153", orc["grey"], ref["grey"])
    same("Canny :162", orc["edges"], ref["edges"])
    names = ["grey", "edges", "median1", "gauss1", "median3", "gauss3", "median5", "gauss5", "median7", "gauss7"]
    for k in range(2, 10):
        same("%s :174-175" % names[k], orc["blurs"][k], ref["blurs"][k])
    for k in range(10):
        same("HoughCircles(%s) :180" % names[k], orc["circles_per_variant"][k], ref["circles_per_variant"][k])
    same("circles :186", orc["circles_all"], ref["circles_all"])
    same("rectangle/circle erase :191-198", orc["circles_removed"], ref["circles_removed"])
    same("HoughLines H :236", orc["hlines"], ref["hlines"])
    same("HoughLines V :240-247", orc["vlines"], ref["vlines"])
    return bad


# ---- digest check: python -m oracle.cv2_harness --digests ---------------------------------------------------------------------

def check_digests(digest_file=None, fixture_dir=None, out=print):
    """Runs the reference's ten calls on the installed cv2 over the committed digest set and compares each stage's digest with
    what oracle/ answered (tests/golden/oracle_stage_digests.json).  Needs numpy, Pillow and cv2 only.  Returns the list of
    (input, stage) that differ under the default switch set; prints the A.7 switch value each directly affected stage matches."""
    import os
    import cv2
    from . import stage_digests as sd
    doc = sd.load(digest_file)
    fixture_dir = fixture_dir or os.path.join(os.path.dirname(sd.DIGEST_FILE), "test_images")
    out("cv2 %s; oracle digests: %d inputs, switches %s" % (cv2.__version__, len(doc["inputs"]), doc["switches"]))
    bad, votes = [], {"grey_shift": {}, "gauss_kernel_mode": {}, "houghlines_numangle": {}}

    def vote(switch, value):
        votes[switch][value] = votes[switch].get(value, 0) + 1

    for name, img in sd.inputs(fixture_dir):
        e = doc["inputs"].get(name)
        if e is None:
            continue
        if sd.sha(img, np.uint8) != e["input"]:
            out("%-16s INPUT differs (another Pillow / libjpeg decodes or enhances these pixels differently): skipped" % name)
            continue
        got = sd.stage_digests(run_cv2_calls(img, e["threshold"]))
        for stage, want in e["stages"].items():
            if got[stage] == want:
                continue
            alts = [k for k, v in e["alt"].items() if k.startswith(stage + "@") and v == got[stage]]
            bad.append((name, stage, alts[0].split("@")[1] if alts else None))
        # which switch value does this cv2 match on the directly affected stages?
        if img.ndim == 3:
            vote("grey_shift", 15 if got["cvtColor:153"] == e["stages"]["cvtColor:153"] else
                 (14 if got["cvtColor:153"] == e["alt"].get("cvtColor:153@grey_shift=14") else "neither"))
        for k in (3, 5, 7):
            st = "gauss%d:174-175" % k
            if got["cvtColor:153"] == e["stages"]["cvtColor:153"]:            # same grey plane into GaussianBlur
                vote("gauss_kernel_mode", 0 if got[st] == e["stages"][st] else (1 if got[st] == e["alt"][st + "@gauss_kernel_mode=1"] else "neither"))
        for st in ("HoughLines_H:236", "HoughLines_V:240-247"):
            if got["erase:191-198"] == e["stages"]["erase:191-198"]:          # same input image to HoughLines
                vote("houghlines_numangle", 0 if got[st] == e["stages"][st] else
                     (1 if got[st] == e["alt"][st + "@houghlines_numangle=1"] else "neither"))
    out("switch values matched (value: number of stages): %s" % votes)
    for name, stage, alt in bad:
        out("DIFFERS %-16s %-28s%s" % (name, stage, "  (matches the oracle under %s)" % alt if alt else ""))
    out("%d of the compared stages differ under the default switch set" % len(bad) if bad else "every compared stage is byte-identical: rows a2-a8 pinned for this cv2")
    return bad


if __name__ == "__main__":
    import argparse
    import sys
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--digests", action="store_true", help="compare the installed cv2 with the committed oracle digests")
    ap.add_argument("--file", default=None)
    ap.add_argument("--fixtures", default=None)
    a = ap.parse_args()
    if a.digests:
        if not have_cv2():
            sys.exit("cv2 is not importable here: nothing to compare")
        sys.exit(1 if check_digests(a.file, a.fixtures) else 0)
    ap.print_help()
