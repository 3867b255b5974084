"""Parity checks shared by the emulated (CPU, tests/emu) and the real (GPU) test suites: run a Detector and the
oracle on the same image and compare every stage bit for bit."""
import numpy as np

from oracle import cv_oracle as cvo
from oracle import pipeline as opipe

VARIANT_PLANES = ["grey", "edges", "median3", "gauss3", "median5", "gauss5", "median7", "gauss7"]


def oracle_variants(ref):
    b = ref["blurs"]
    return [b[0], b[1], b[4], b[5], b[6], b[7], b[8], b[9]]


def compare_planes(det, index, ref):
    """grey, main Canny, blur bank, circles_removed."""
    for name, want in zip(VARIANT_PLANES, oracle_variants(ref)):
        got = det.fetch_plane(index, name)
        bad = np.argwhere(got != want)
        assert len(bad) == 0, "%s differs at %d px, first %s got %d want %d" % (
            name, len(bad), bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
    got = det.fetch_plane(index, "removed")
    bad = np.argwhere(got != ref["circles_removed"])
    assert len(bad) == 0, "circles_removed differs at %d px, first %s" % (len(bad), bad[0])


def compare_hough_internals(det, index, ref, variants=range(8)):
    """HoughCircles-internal Canny maps and vote accumulators per variant (needs det.set_debug(True))."""
    ov = oracle_variants(ref)
    for v in variants:
        _, dbg = cvo.hough_circles(ov[v], debug=True)
        m = det.fetch_plane(index, 9 + 1 + v)
        got_edges = (m == 2).astype(np.uint8) * 255
        bad = np.argwhere(got_edges != dbg["edges"])
        assert len(bad) == 0, "variant %d hough-canny differs at %d px, first %s" % (v, len(bad), bad[0])
        acc = det.fetch_circle_acc(index, v)
        h, w = acc.shape
        want = dbg["acc"][:h, :w]
        bad = np.argwhere(acc != want)
        assert len(bad) == 0, "variant %d accumulator differs at %d cells, first %s got %d want %d" % (
            v, len(bad), bad[0], acc[tuple(bad[0])], want[tuple(bad[0])])


def compare_detection(d, ref):
    """Detection (product) against the oracle's process_image dict: exact."""
    per = ref["circles_per_variant"]
    assert d.n_per_slot == [len(c) for c in per], (d.n_per_slot, [len(c) for c in per])
    np.testing.assert_array_equal(d.circles_all, ref["circles_all"])
    assert d.threshold == ref["threshold"]
    np.testing.assert_array_equal(d.hlines, ref["hlines"])
    np.testing.assert_array_equal(d.vlines, ref["vlines"])
    np.testing.assert_array_equal(d.hcentres, ref["hcentres"])
    np.testing.assert_array_equal(d.vcentres, ref["vcentres"])
    assert d.found_grid == ref["found_grid"] and d.valid_grid == ref["valid_grid"]
    assert d.board_ready == ref["board_ready"]
    assert (d.hsize, d.vsize) == (ref["hsize"], ref["vsize"])
    if ref["valid_grid"]:
        np.testing.assert_array_equal(d.hcentres_complete, ref["hcentres_complete"])
        np.testing.assert_array_equal(d.vcentres_complete, ref["vcentres_complete"])
        assert d.hspace == ref["hspace"] and d.vspace == ref["vspace"]
    np.testing.assert_array_equal(d.circles, ref["circles"])
    if ref["board_ready"]:
        np.testing.assert_array_equal(d.detected_board, ref["detected_board"])
        np.testing.assert_array_equal(d.full_board, ref["full_board"])
        np.testing.assert_array_equal(d.stone_brightnesses, ref["stone_brightnesses"])
        assert (d.num_black_stones, d.num_white_stones, d.side_to_move) == (
            ref["num_black_stones"], ref["num_white_stones"], ref["side_to_move"])
        assert d.sgf == ref["sgf"]
    else:
        assert d.sgf is None


def run_and_compare(det, images, params=None, internals=False, oracle_kwargs=None):
    """Full check of a batch; returns the Detections."""
    if internals:
        det.set_debug(True)
    dets = det.detect_batch(images, params, full=True)
    nb_last = (len(images) - 1) % det.max_batch + 1
    first_last = len(images) - nb_last
    for k, (img, d) in enumerate(zip(images, dets)):
        ref = opipe.process_image(img, **(oracle_kwargs or {}))
        if k >= first_last:                    # planes of the last device pass are still resident
            compare_planes(det, k - first_last, ref)
            if internals:
                compare_hough_internals(det, k - first_last, ref)
        compare_detection(d, ref)
    return dets
