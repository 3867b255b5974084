"""The OpenCV-version question: tests/golden/fixture_outcomes.json holds what the oracle makes
of the reference's 18 fixtures at the reference's default settings under every named switch set (tests/switches.py).

CPU: the file is what the oracle answers today (a subset here, all of it with I2S_ALL_OUTCOMES=1), it records the evidence the
default was chosen on, Params.opencv_switches maps releases to sets, and the emulated kernels reproduce non-default sets.
GPU: the HIP path reproduces EVERY entry (18 fixtures x 6 switch sets) through the C ABI."""
import json
import os
import sys

import numpy as np
import pytest

import switches
from helpers import GOLDEN
from img2sgf_amd.pipeline import Detector, Params, board_to_sgf
from oracle import pipeline as opipe

sys.path.insert(0, GOLDEN)
import make_fixture_outcomes as mk  # noqa: E402

with open(mk.OUT) as f:
    DOC = json.load(f)


def _img(name):
    return opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", name))


def test_file_covers_every_fixture_and_switch_set():
    assert DOC["switch_sets"] == {s: switches.compat(s) for s in switches.NAMES}
    assert sorted(DOC["fixtures"]) == sorted(mk.IMAGES)
    for per in DOC["fixtures"].values():
        assert sorted(per) == sorted(switches.NAMES)
    # every value of every switch is exercised by some set
    for key, values in (("houghlines_numangle", {0, 1}), ("grey_shift", {14, 15}), ("gauss_kernel_mode", {0, 1})):
        assert {s[key] for s in switches.SWITCH_SETS.values()} == values


def test_oracle_reproduces_the_file():
    names = mk.IMAGES if os.environ.get("I2S_ALL_OUTCOMES") else ["ex1.jpg", "ex8.jpg", "ex9.jpg", "ex10.jpg", "no_circles.jpg"]
    fresh = mk.build(names)
    for n in names:
        assert fresh["fixtures"][n] == DOC["fixtures"][n], n


def test_defaults_are_the_documented_switch_set():
    """Product defaults == oracle defaults == the "numangle_legacy" set (OpenCV 4.3 .. 4.5.1), in Python and in the C ABI."""
    from img2sgf_amd import _lib
    from oracle import cv_oracle as cvo
    assert cvo.DEFAULT_COMPAT == switches.compat("numangle_legacy")
    assert Params().switch_set() == switches.params_kwargs("numangle_legacy")
    p = _lib.I2sParams()
    _lib.load().dll.i2s_default_params(__import__("ctypes").byref(p))
    assert (p.houghlines_numangle_mode, p.grey_shift, p.gauss_kernel_mode) == (1, 15, 0)


def test_opencv_release_to_switch_set():
    f = Params.opencv_switches
    assert f("4.2.0") == f("4.2.0.34") == switches.params_kwargs("opencv_4_2")
    assert f("4.3.0") == f("4.5.1") == f("4.5.1.48") == switches.params_kwargs("numangle_legacy") == Params().switch_set()
    assert f("4.5.2") == f("4.8.1.78") == f("4.10.0") == f("5.0.0-pre") == switches.params_kwargs("current")
    assert f("3.4.9")["grey_shift"] == 14 and f("3.4.9")["gauss_kernel_mode"] == 1
    assert Params.for_opencv("4.8.0", black_threshold=99).houghlines_numangle_mode == 0


def test_evidence_for_the_default():
    """What DESIGN.md 2a argues from: under the angle count of OpenCV <= 4.5.1 the reference author's own clean fixtures give
    full boards, under the current count they do not; the one recorded answer (ex1, screenshot.jpg) is the same under all sets."""
    fx = DOC["fixtures"]
    for s in switches.NAMES:
        assert fx["ex1.jpg"][s]["sgf"] == "(;GM[1]FF[4]SZ[19]\nPL[W]\nAW[cn][jq][nq][qf][qj]\nAB[co][dd][dp][fp][nd][pd][pn][pp][ql]\n)\n"
    legacy, current = "numangle_legacy", "current"
    assert (fx["ex8.jpg"][legacy]["hsize"], fx["ex8.jpg"][legacy]["vsize"]) == (19, 19)       # "clean computer diagram"
    assert (fx["ex8.jpg"][current]["hsize"], fx["ex8.jpg"][current]["vsize"]) == (17, 19)
    assert (fx["ex10.jpg"][legacy]["hsize"], fx["ex10.jpg"][legacy]["vsize"]) == (19, 8)      # top side, 19 wide x ~8 tall
    assert (fx["ex10.jpg"][current]["hsize"], fx["ex10.jpg"][current]["vsize"]) == (19, 3)
    assert fx["ex16.jpg"][legacy]["board_ready"] and not fx["ex16.jpg"][current]["board_ready"]
    boards = {s: sum(1 for n in mk.IMAGES if fx[n][s]["board_ready"]) for s in switches.NAMES}
    assert boards[legacy] == 15 and boards[current] == 14


@pytest.mark.parametrize("sw", ["numangle_legacy", "current", "all_alternative"])
def test_emulated_kernels_under_switch_sets(sw):
    """The unmodified kernels under the CPU emulation: a colour-free part board (ex9), the top-side board whose outcome depends on
    the angle count (ex10), every stage against the oracle under the same switch set, and the outcome against the file."""
    import emu_util
    import parity
    lib = emu_util.emu_library()
    for name in ("ex9.jpg", "ex10.jpg"):
        img = _img(name)
        det = Detector(0, 1, img.shape[1], img.shape[0], lib=lib)
        d = parity.run_and_compare(det, [img], params=switches.params(sw), oracle_kwargs=dict(compat=switches.compat(sw)))[0]
        det.close()
        want = DOC["fixtures"][name][sw]
        assert (d.board_ready, d.hsize, d.vsize, d.sgf) == (want["board_ready"], want["hsize"], want["vsize"], want["sgf"])


@pytest.mark.gpu
@pytest.mark.parametrize("sw", switches.NAMES)
def test_hip_path_reproduces_every_outcome(sw):
    """BASELINE configs[4] under every switch set: all 18 fixtures as one ragged batch through the C ABI; board size, counts,
    side to move and SGF bytes equal the committed outcome."""
    imgs = [_img(n) for n in mk.IMAGES]
    det = Detector(0, 6, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
    dets = det.detect_batch(imgs, switches.params(sw), full=True)
    boards = det.detect_batch(imgs, switches.params(sw), full=False)
    det.close()
    for n, d, b in zip(mk.IMAGES, dets, boards):
        want = DOC["fixtures"][n][sw]
        got = dict(threshold=d.threshold, n_circles=len(d.circles_all), n_hlines=len(d.hlines), n_vlines=len(d.vlines),
                   n_hclusters=len(d.hcentres), n_vclusters=len(d.vcentres), found_grid=d.found_grid, valid_grid=d.valid_grid,
                   board_ready=d.board_ready, hsize=d.hsize, vsize=d.vsize, sgf=d.sgf)
        if d.board_ready:
            got.update(n_black=d.num_black_stones, n_white=d.num_white_stones, side_to_move=d.side_to_move)
        assert got == want, (n, sw)
        assert (board_to_sgf(b) if b.status == 0 else None) == want["sgf"], (n, sw)
