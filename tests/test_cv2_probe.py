"""the OpenCV switch set is selected by what a live cv2 module COMPUTES (pipeline.probe_cv2_switches,
Params.from_cv2, gui_adapter.install), not by its version string.  cv2 is absent here, so the module under the probe is the
oracle-backed stand-in of tests/test_cv2_harness_plumbing.py -- a real module object whose ten calls answer with the oracle under a chosen
switch set.  That also checks the probes' closed forms (impulse / line responses of the two-pass 8.8 fixed-point Gaussian, one-pixel
HoughLines counts, the two grey coefficient sets) against the oracle's full restatement of each call.  Says nothing about OpenCV itself."""
import itertools
import types
from unittest.mock import MagicMock

import numpy as np
import pytest

import switches
from img2sgf_amd import gui_adapter, pipeline
from img2sgf_amd.pipeline import I2sError, Params
from test_cv2_harness_plumbing import _oracle_backed_module


def _standin(sw):
    return _oracle_backed_module(sw["grey_shift"], sw["gauss_kernel_mode"], sw["houghlines_numangle_mode"])


@pytest.mark.parametrize("name", switches.NAMES)
def test_probe_picks_exactly_the_standins_set(name):
    sw = switches.params_kwargs(name)
    assert pipeline.probe_cv2_switches(_standin(sw)) == sw
    assert Params.from_cv2(_standin(sw), line_threshold=80).switch_set() == sw
    assert Params.from_cv2(_standin(sw), line_threshold=80).line_threshold == 80


def test_probe_all_eight_combinations():
    for gs, gm, na in itertools.product((15, 14), (0, 1), (0, 1)):
        assert pipeline.probe_cv2_switches(_oracle_backed_module(gs, gm, na)) == dict(
            grey_shift=gs, gauss_kernel_mode=gm, houghlines_numangle_mode=na)


def test_unrecognised_behaviour_raises_instead_of_guessing():
    m = _oracle_backed_module()
    real = m.cvtColor
    m.cvtColor = lambda img, code: np.minimum(real(img, code).astype(int) + 1, 255).astype(np.uint8)      # some third coefficient set
    with pytest.raises(I2sError, match="cvtColor"):
        pipeline.probe_cv2_switches(m)
    m = _oracle_backed_module()
    real_g = m.GaussianBlur
    m.GaussianBlur = lambda g, ks, s: real_g(g, ks, s) // 2 * 2                                          # not the fixed-point path
    with pytest.raises(I2sError, match="GaussianBlur"):
        pipeline.probe_cv2_switches(m)
    m = _oracle_backed_module()
    real_h = m.HoughLines

    def four(img, rho, theta, threshold, min_theta, max_theta):
        r = real_h(img, rho, theta, threshold, min_theta, max_theta)
        return None if r is None else np.concatenate([r, r[:1]])                                          # one angle too many
    m.HoughLines = four
    with pytest.raises(I2sError, match="HoughLines"):
        pipeline.probe_cv2_switches(m)
    m = _oracle_backed_module()
    m.HoughLines = lambda *a, **k: None
    with pytest.raises(I2sError, match="HoughLines"):
        pipeline.probe_cv2_switches(m)


class _Var:
    def __init__(self, v):
        self.v = v

    def get(self):
        return self.v


def _fake_reference_module(cv):
    m = types.SimpleNamespace()
    m.cv = cv
    m.image_loaded = True
    m.edge_min, m.edge_max, m.threshold, m.sobel, m.gradient = _Var(50), _Var(200), _Var(80), _Var(3), _Var(1)
    m.black_stone_threshold, m.board_alignment = 128, (2, 0)
    return m


@pytest.mark.parametrize("name", switches.NAMES)
def test_gui_adapter_reads_the_switches_off_the_live_module(name):
    """The version string of the stand-in ("0.0-oracle-...") would select the 3.x / oldest set: the adapter must not look at it."""
    sw = switches.params_kwargs(name)
    st = gui_adapter.install(_fake_reference_module(_standin(sw)))
    assert st["switches"] == sw and st["switches_from"] == "probed"


def test_gui_adapter_fallbacks():
    mock = MagicMock()                                   # the headless tests' cv2: not a module object, never probed
    mock.__version__ = "4.8.1"
    st = gui_adapter.install(_fake_reference_module(mock))
    assert st["switches"] == Params.opencv_switches("4.8.1") and st["switches_from"] == "version string"
    st = gui_adapter.install(_fake_reference_module(MagicMock()))
    assert st["switches"] == {} and st["switches_from"] == "package defaults"
    st = gui_adapter.install(_fake_reference_module(MagicMock()), opencv="4.2.0")
    assert st["switches"] == Params.opencv_switches("4.2.0") and st["switches_from"] == "given"
    st = gui_adapter.install(_fake_reference_module(MagicMock()), opencv=switches.params_kwargs("grey14"))
    assert st["switches"] == switches.params_kwargs("grey14")
    # a live module that answers strangely: Params.from_cv2 raises; the adapter falls back to the version string and SAYS so
    bad = _oracle_backed_module()
    bad.HoughLines = lambda *a, **k: None
    bad.__version__ = "4.5.1"
    with pytest.raises(I2sError):
        Params.from_cv2(bad)
    m = _fake_reference_module(bad)
    said = []
    m.log = said.append
    st = gui_adapter.install(m)
    assert st["switches"] == Params.opencv_switches("4.5.1") and st["switches_from"].startswith("version string (the probe of the live module failed")
    assert len(said) == 1 and "HoughLines" in said[0]
    # ... also when a call of the module raises instead of answering
    def boom(*a, **k):
        raise RuntimeError("threshold must be positive")
    bad.HoughLines = boom
    st = gui_adapter.install(_fake_reference_module(bad))
    assert "RuntimeError" in st["switches_from"] and st["switches"] == Params.opencv_switches("4.5.1")


@pytest.mark.parametrize("sobel,gradient,what", [(5, 1, "apertureSize"), (7, 1, "apertureSize"), (3, 2, "L2gradient")])
def test_gui_adapter_refuses_canny_flavours_it_does_not_implement(sobel, gradient, what):
    """cv.Canny(.., apertureSize=sobel.get(), L2gradient=(gradient.get()==2)), img2sgf.py:164-165: the widgets are hidden (:1142-1182),
    every user runs 3 / L1 -- a patched application that sets them must get an error, not a silently different answer (VERDICT r4 missing 6)."""
    m = _fake_reference_module(MagicMock())
    gui_adapter.install(m)
    m.sobel, m.gradient = _Var(sobel), _Var(gradient)
    with pytest.raises(I2sError, match=what):
        m.process_image()
    with pytest.raises(I2sError, match=what):
        m.identify_board()
    m.image_loaded = False
    m.process_image()                                    # the reference's early return (:118) comes first


def test_default_params_hint_when_the_imported_cv2_differs(monkeypatch):
    """pipeline.process_image on the package defaults, in a process that has imported a cv2 whose arithmetic is another set: one
    warning that names both sets and Params.from_cv2; none when the sets agree, when Params are given, or without a cv2."""
    import sys
    import warnings
    calls = []
    monkeypatch.setattr(pipeline, "_detector_for", lambda images, detector: (_ for _ in ()).throw(RuntimeError("stop here")))

    def run(params=None):
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            with pytest.raises(RuntimeError, match="stop here"):
                pipeline.process_image(np.zeros((8, 8), np.uint8), params)
        return [str(x.message) for x in w]

    monkeypatch.setattr(pipeline, "_HINTED", False)
    monkeypatch.delitem(sys.modules, "cv2", raising=False)
    assert run() == []
    monkeypatch.setitem(sys.modules, "cv2", _standin(switches.params_kwargs("current")))
    monkeypatch.setattr(pipeline, "_HINTED", False)
    assert run(Params()) == []                                   # explicit parameters: the caller has decided
    msgs = run()
    assert len(msgs) == 1 and "from_cv2" in msgs[0] and "'houghlines_numangle_mode': 0" in msgs[0]
    assert run() == []                                           # once per process
    monkeypatch.setitem(sys.modules, "cv2", _standin(Params().switch_set()))
    monkeypatch.setattr(pipeline, "_HINTED", False)
    assert run() == []                                           # the installed module computes what the defaults restate
