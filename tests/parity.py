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


def check_validate_grid_capacity(det, trials):
    """i2s_validate_grid (img2sgf.py:420-445) on 0, 1, 2 ... 1024 centres per direction -- the record's capacity, which is the line
    capacity -- uniform, with missing lines, closer than min_grid_spacing, with jitter: the reference's eight outputs, exactly."""
    from img2sgf_amd import pipeline
    from img2sgf_amd.pipeline import Params
    from oracle import glue
    rng = np.random.default_rng(3)

    def centres(n):
        if n == 0:
            return np.zeros(0)
        sp = rng.choice([3.0, 9.99, 10.0, 12.5, 30.0])
        return np.cumsum(np.full(n, sp) + (rng.random(n) < 0.15) * sp * rng.integers(1, 4, n) + rng.random(n) * rng.choice([0, 0.5, 3]))

    for trial in range(trials):
        hc = centres(int(rng.choice([0, 1, 2, 5, 19, 20, 21, 22, 40, 300, 1024])))
        vc = centres(int(rng.choice([0, 1, 2, 19, 21, 23, 257, 1000, 1024])))
        circles = np.stack([rng.uniform(0, 600, 30), rng.uniform(0, 600, 30), rng.uniform(1, 40, 30)], 1).astype(np.float32)
        got = pipeline.validate_grid(hc, vc, circles, Params(), det)
        want = glue.validate_grid(hc, vc, circles)
        assert bool(got[0]) == bool(want["valid"]), trial
        if want["valid"]:
            assert (got[2], got[3], got[6], got[7]) == (want["vsize"], want["hsize"], want["hspace"], want["vspace"]), trial
            np.testing.assert_array_equal(got[4], want["hc"])
            np.testing.assert_array_equal(got[5], want["vc"])
            np.testing.assert_array_equal(np.asarray(got[1], np.float32).reshape(-1, 3), np.asarray(want["circles"], np.float32).reshape(-1, 3))


def check_find_lines_degenerate(det, size):
    """find_all_lines (img2sgf.py:230-265, three cv.HoughLines calls) on injected images at the edges: all white (every pixel votes),
    all black, one pixel, stripes of period 2 .. 11 in both directions (hundreds of peaks), a checkerboard, a grid, noise, 1 x N and
    N x 1 images, grey values other than 0 / 255 (any non-zero pixel votes) -- thresholds from 1 vote up: the rho lists in the
    reference's output order, exactly; a direction with more than I2S_MAX_LINES peaks must be refused, not cut.  size = the largest side."""
    from img2sgf_amd import pipeline
    from img2sgf_amd.pipeline import I2sError, Params
    from oracle import glue
    rng = np.random.default_rng(4)
    s = size
    cases = [np.full((3 * s // 7, 4 * s // 7), 255, np.uint8), np.zeros((3 * s // 7, 4 * s // 7), np.uint8),
             np.pad(np.full((1, 1), 255, np.uint8), ((s // 7, s // 14), (s // 35, 3 * s // 7)))]
    for per in (2, 3, 5, 9, 11):
        a = np.zeros((s, s - 10), np.uint8)
        a[::per] = 255
        cases += [a, np.ascontiguousarray(a.T[:s - 50])]
    cases += [((np.indices((s // 2 - 17, 4 * s // 7 + 1)).sum(0) % 2) * 255).astype(np.uint8),
              np.maximum(*[(np.indices((6 * s // 7, 6 * s // 7))[k] % 31 == 0) * 255 for k in (0, 1)]).astype(np.uint8),
              (rng.random((5 * s // 7, 9 * s // 10)) < 0.1).astype(np.uint8) * 255, np.full((1, s), 255, np.uint8), np.full((s, 1), 255, np.uint8),
              rng.integers(0, 256, (2 * s // 7, 3 * s // 7), dtype=np.uint8)]
    flat = lambda a: np.asarray(a, np.float32).reshape(-1)
    for n, im in enumerate(cases):
        for thr in (1, 2, 20, 74, 300):
            want_h, want_v = glue.find_lines(im, thr, True), glue.find_lines(im, thr, False)
            try:
                got_h, got_v = pipeline.find_all_lines(im, thr, Params(), det)
            except I2sError:
                assert len(want_h) > 1024 or len(want_v) > 1024, (n, thr)
                continue
            np.testing.assert_array_equal(flat(got_h), flat(want_h), err_msg="case %d threshold %d" % (n, thr))
            np.testing.assert_array_equal(flat(got_v), flat(want_v), err_msg="case %d threshold %d" % (n, thr))


def lines_overflow_image(seed=0, w=3000, h=160):
    """A sparse-noise strip with four discs: at a Hough-line threshold of a few votes it has far more than I2S_MAX_LINES peaks in a
    direction (status CAPACITY out of k_line_peaks) while HoughCircles has found circles on it -- ADVICE r4's reproduction."""
    rng = np.random.default_rng(seed)
    img = np.where(rng.random((h, w)) < 0.012, 0, 255).astype(np.uint8)
    yy, xx = np.mgrid[:h, :w]
    for k in range(4):
        img[(xx - (300 + 700 * k)) ** 2 + (yy - 80) ** 2 <= 20 ** 2] = 0
    return img


def check_capacity_by_lines_in_full_batch(det, diagrams, threshold=8):
    """an image whose line peaks overflow (status CAPACITY, circles present) placed FIRST and LAST in a full=True
    batch.  Its packed full record must not overrun its slot: the neighbours' records equal the oracle's, its own says "nothing
    valid" in every count and carries no board of an earlier pass.  `diagrams`: images that detect normally."""
    from img2sgf_amd.pipeline import Params, _detection_from_result
    strip = lines_overflow_image()
    diagrams = list(diagrams)
    batch = [strip] + diagrams + [strip]
    params = Params(line_threshold=threshold)
    ref = opipe.process_image(strip, threshold=threshold, keep_planes=False)
    assert len(ref["hlines"]) > 1024 or len(ref["vlines"]) > 1024, "the strip no longer overflows the line capacity"
    assert len(ref["circles_all"]) > 0, "the strip must carry circles for this test to mean anything"
    # an earlier pass leaves boards in every slot of the context
    det.detect_batch((diagrams * len(batch))[:len(batch)], full=True)
    imgs = [np.ascontiguousarray(im) for im in batch]
    boards, res = det.detect_ptrs([im.ctypes.data for im in imgs], [im.shape[1] for im in imgs], [im.shape[0] for im in imgs],
                                  [im.strides[0] for im in imgs], [1] * len(imgs), params, False, True)
    for k in (0, len(batch) - 1):
        r, b = res[k], boards[k]
        assert r.status == 100 and b.status == 100 and not r.board_ready
        assert r.n_circles == 0 and b.n_circles == 0 and r.n_circles_kept == 0 and r.n_stones == 0
        assert r.n_hlines == 0 and r.n_vlines == 0
        assert not np.ctypeslib.as_array(r.board).any() and not np.ctypeslib.as_array(r.detected).any()
        assert not np.ctypeslib.as_array(b.board).any()
        assert _detection_from_result(r).sgf is None
    for k in range(1, len(batch) - 1):
        compare_detection(_detection_from_result(res[k]), opipe.process_image(batch[k], threshold=threshold, keep_planes=False))
