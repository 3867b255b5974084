"""Cross-check of the oracle -- and, on a GPU box, of the HIP path -- against a REAL OpenCV (SURVEY 8c last bullet).

Dormant where cv2 is absent (this image and today's GPU boxes): the whole module skips.  Where `import cv2` works it
(1) prints cv2.__version__ (the reference logs it, img2sgf.py:1246), (2) auto-selects the version switches of SURVEY
Appendix A.7, (3) byte-compares every one of the ten OpenCV calls of img2sgf.py:153, 162-165, 174, 175, 180,
197-198, 236-244 between cv2 and oracle/ on the 18 reference fixtures and three synthetic diagrams, and (4) compares the
SGF the reference would write.  Green here turns rows a2-a8 from "parity unpinned" into pinned.
"""
import glob
import os

import numpy as np
import pytest

cv2 = pytest.importorskip("cv2")

from helpers import GOLDEN                       # noqa: E402
from img2sgf_amd import synth                    # noqa: E402
from oracle import cv2_harness as H              # noqa: E402
from oracle import pipeline as opipe             # noqa: E402

FIXTURES = sorted(glob.glob(os.path.join(GOLDEN, "test_images", "*.jpg")))


@pytest.fixture(scope="module")
def compat():
    c = H.select_compat()
    print("\ncv2 %s: oracle switches %s" % (cv2.__version__, c))
    return c


def _inputs():
    for p in FIXTURES:
        yield os.path.basename(p), opipe.load_and_enhance(p)            # the reference's defaults: contrast 70, brightness 50
    for seed in (0, 1):
        yield "synth%d" % seed, synth.synth_diagram(seed)[0]
    yield "synth_noisy0", synth.synth_diagram(0, noisy=True)[0]


def test_version_and_switches(compat):
    assert set(compat) == {"grey_shift", "gauss_kernel_mode", "houghlines_numangle"}


def test_product_probe_agrees_with_the_harness(compat):
    """The package's own behavioural probe (pipeline.probe_cv2_switches: three closed-form probes, no oracle involved) on the live cv2
    against the harness's selection (the oracle's kernels compared with cv2 on probe images): two independent readings of the same
    module must name the same switch set -- until now the probe has only ever met the oracle-backed stand-in (tests/test_cv2_probe.py).
    Also: the release table's entry for this version (from memory, pipeline.Params.opencv_switches) is checked against both."""
    from img2sgf_amd import pipeline
    got = pipeline.probe_cv2_switches(cv2)
    want = dict(grey_shift=compat["grey_shift"], gauss_kernel_mode=compat["gauss_kernel_mode"], houghlines_numangle_mode=compat["houghlines_numangle"])
    assert got == want, "cv2 %s: probe %s, harness %s" % (cv2.__version__, got, want)
    assert pipeline.Params.opencv_switches(cv2.__version__) == want, "the release table is wrong for cv2 %s: %s" % (cv2.__version__, want)


@pytest.mark.parametrize("name,img", list(_inputs()), ids=lambda v: v if isinstance(v, str) else "")
def test_ten_calls_bytewise(name, img, compat):
    bad = H.compare(img, compat)
    assert not bad, "%s, cv2 %s: %s" % (name, cv2.__version__, bad)


@pytest.mark.parametrize("path", FIXTURES, ids=os.path.basename)
def test_sgf_matches_cv2_run(path, compat):
    img = opipe.load_and_enhance(path)
    ref = H.cv2_process_image(img)
    orc = opipe.process_image(img, compat=compat, keep_planes=False)
    assert orc["board_ready"] == ref["board_ready"]
    assert orc["sgf"] == ref["sgf"]


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=os.path.basename)
def test_hip_path_matches_cv2_run(path, compat):
    """BASELINE.json's acceptance line: identical 19x19 matrix / byte-identical SGF, circles within +-1 px of cv2's."""
    from img2sgf_amd.pipeline import Detector, Params
    img = opipe.load_and_enhance(path)
    ref = H.cv2_process_image(img)
    det = Detector(0, 1, img.shape[1], img.shape[0])
    d = det.detect_batch([img], Params(grey_shift=compat["grey_shift"], gauss_kernel_mode=compat["gauss_kernel_mode"],
                                       houghlines_numangle_mode=compat["houghlines_numangle"]))[0]
    det.close()
    assert d.board_ready == ref["board_ready"]
    assert d.sgf == ref["sgf"]
    assert len(d.circles_all) == len(ref["circles_all"])
    if len(d.circles_all):
        assert np.abs(d.circles_all - ref["circles_all"]).max() <= 1.0
