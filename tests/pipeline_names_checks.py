"""Checks of the stateless reference-named entry points of img2sgf_amd.pipeline (find_lines / find_all_lines 230-265,
cluster_lines 295-332, validate_grid 420-445), shared by the emulated (CPU) and the real (GPU) test modules.  Expected values:
the golden vectors generated from the reference itself (tests/golden/glue_golden.json) for the glue, the oracle for HoughLines."""
import numpy as np

from helpers import load_glue_golden
from img2sgf_amd import pipeline, synth
from oracle import glue
from oracle import pipeline as opipe

G = load_glue_golden()


def check_cluster_lines(det):
    for entry in G["cases"]:
        c, e = entry["case"], entry["expect"]
        hc, vc, found = pipeline.cluster_lines(np.array(c["hlines"], np.float32), np.array(c["vlines"], np.float32), detector=det)
        np.testing.assert_array_equal(hc, np.asarray(e["hcentres"], np.float64), err_msg=c["name"])
        np.testing.assert_array_equal(vc, np.asarray(e["vcentres"], np.float64), err_msg=c["name"])
        assert found == e["found_grid"], c["name"]


def check_validate_grid(det):
    n = 0
    for entry in G["cases"]:
        c, e = entry["case"], entry["expect"]
        if not e["found_grid"]:
            continue
        circles = np.array(c["circles"], np.float32).reshape(-1, 3)
        out = pipeline.validate_grid(e["hcentres"], e["vcentres"], circles, detector=det)
        assert out[0] == e["valid_grid"], c["name"]
        if e["valid_grid"]:
            valid, kept, vsize, hsize, hcc, vcc, hspace, vspace = out
            assert (hsize, vsize) == (e["hsize"], e["vsize"]), c["name"]
            np.testing.assert_array_equal(hcc, np.asarray(e["hcentres_complete"], np.float64), err_msg=c["name"])
            np.testing.assert_array_equal(vcc, np.asarray(e["vcentres_complete"], np.float64), err_msg=c["name"])
            assert (hspace, vspace) == (e["hspace"], e["vspace"]), c["name"]
            np.testing.assert_array_equal(np.asarray(kept, np.float32).reshape(-1, 3),
                                          np.asarray(e["kept_circles"], np.float32).reshape(-1, 3), err_msg=c["name"])
            n += 1
        else:
            assert out[2:] == (0, 0, None, None, None, None)
    assert n >= 10


def check_find_lines(det):
    img = synth.synth_diagram(2, geom=synth.GEOM_SMALL)[0]
    ref = opipe.process_image(img)
    removed, thr = ref["circles_removed"], ref["threshold"]
    hl, vl = pipeline.find_all_lines(removed, thr, detector=det)
    want_h, want_v = glue.find_lines(removed, thr, True), glue.find_lines(removed, thr, False)
    np.testing.assert_array_equal(np.asarray(hl, np.float32).reshape(-1), np.asarray(want_h, np.float32).reshape(-1))
    np.testing.assert_array_equal(np.asarray(vl, np.float32).reshape(-1), np.asarray(want_v, np.float32).reshape(-1))
    assert np.asarray(hl).shape == (len(ref["hlines"]), 1)                    # the (n,1) column find_lines returns (:255)
    np.testing.assert_array_equal(pipeline.find_lines(removed, thr, pipeline.HORIZONTAL, detector=det), hl)
    np.testing.assert_array_equal(pipeline.find_lines(removed, thr, pipeline.VERTICAL, detector=det), vl)
    # nothing above the threshold: [] like the reference (:255)
    empty_h, empty_v = pipeline.find_all_lines(np.zeros((40, 50), np.uint8), 20, detector=det)
    assert len(empty_h) == 0 and len(empty_v) == 0
    # a non-default angle tolerance changes the angle set of the three HoughLines calls
    p = pipeline.Params(angle_tolerance=1.4)
    hl2, _ = pipeline.find_all_lines(removed, thr, params=p, detector=det)
    import math
    from oracle import cv_oracle as cvo
    dlt = math.pi / 180 * 1.4
    want = cvo.hough_lines(removed, 1, math.pi / 180.0, thr, math.pi / 2 - dlt, math.pi / 2 + dlt)
    want = np.zeros(0, np.float32) if want is None else want[:, 0, 0]
    np.testing.assert_array_equal(np.asarray(hl2, np.float32).reshape(-1), want)
