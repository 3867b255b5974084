"""SURVEY 8f-3 on the GPU box: img2sgf_amd.gui_adapter.install() on the REAL library.  /root/reference does not exist there,
so the reference module is replaced by a plain namespace that carries exactly the globals and `.get()` holders install()
touches (no reference code travels); the consumers the adapter feeds are draw_images (img2sgf.py:862-897), draw_board
(:900-952), draw_histogram (:207-227) and apply_black_thresh (:762-766) -- the test asserts that every global they read
holds what the oracle computes for the same input, and that the SGF written from `full_board` is the golden one."""
import os
import types

import numpy as np
import pytest
from PIL import Image

from helpers import GOLDEN
from img2sgf_amd import gui_adapter, pipeline, preprocess
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu


class Var:
    def __init__(self, v):
        self.v = v

    def get(self):
        return self.v

    def set(self, v):
        self.v = v


def stand_in(path):
    """What the reference module looks like to install() after open_file() + initialise_parameters() (img2sgf.py:616-660)."""
    m = types.SimpleNamespace()
    m.calls, m.logged = [], []
    m.log = lambda msg: m.logged.append(msg)
    m.draw_board = lambda *a: m.calls.append("draw_board")
    m.draw_images = lambda *a: m.calls.append("draw_images")
    m.draw_histogram = lambda *a: m.calls.append("draw_histogram")
    m.input_image_PIL = Image.open(path).convert("RGB")                       # :651
    w, h = m.input_image_PIL.size
    m.image_loaded = True
    m.found_grid = m.valid_grid = m.board_ready = False
    m.rotate_angle, m.contrast, m.brightness = Var(0), Var(70), Var(50)      # :629-632
    m.black_stone_threshold = 128                                             # :633
    m.board_alignment = [pipeline.LEFT, pipeline.TOP]                         # :627
    m.selection_global = [0, 0, w, h]                                         # :636
    m.threshold = Var(pipeline.choose_threshold((w, h)))                      # :638
    m.edge_min, m.edge_max = Var(50), Var(200)
    m.side_to_move = Var(1)
    m.stone_brightnesses, m.num_black_stones, m.num_white_stones = [], 0, 0
    m.tk = types.SimpleNamespace(ACTIVE="active")
    m.save_button = types.SimpleNamespace(state=None)
    m.save_button.configure = lambda state=None: setattr(m.save_button, "state", state)
    return m


def check_against_oracle(m, ref):
    np.testing.assert_array_equal(m.input_image_np, ref["input"])
    np.testing.assert_array_equal(np.array(m.region_PIL), ref["input"])                       # draw_images :866
    np.testing.assert_array_equal(m.grey_image_np, ref["grey"])
    np.testing.assert_array_equal(m.edge_detected_image_np, ref["edges"])                      # :880
    np.testing.assert_array_equal(np.array(m.edge_detected_image_PIL), ref["edges"])
    np.testing.assert_array_equal(m.circles_removed_image_np, ref["circles_removed"])          # :884
    assert (bool(m.found_grid), bool(m.valid_grid), bool(m.board_ready)) == (ref["found_grid"], ref["valid_grid"], ref["board_ready"])
    np.testing.assert_array_equal(np.asarray(m.circles, np.float32).reshape(-1, 3), ref["circles"])   # :874-878
    np.testing.assert_array_equal(m.hcentres, ref["hcentres"])                                  # :887-891
    np.testing.assert_array_equal(m.vcentres, ref["vcentres"])
    if ref["valid_grid"]:
        np.testing.assert_array_equal(m.hcentres_complete, ref["hcentres_complete"])            # :892-896
        np.testing.assert_array_equal(m.vcentres_complete, ref["vcentres_complete"])
        assert (m.hsize, m.vsize, m.hspace, m.vspace) == (ref["hsize"], ref["vsize"], ref["hspace"], ref["vspace"])
    if ref["board_ready"]:
        np.testing.assert_array_equal(m.full_board, ref["full_board"])                          # draw_board :900-952
        np.testing.assert_array_equal(m.detected_board, ref["detected_board"])
        np.testing.assert_array_equal(m.stone_brightnesses, ref["stone_brightnesses"])          # draw_histogram :207-227
        assert (m.num_black_stones, m.num_white_stones) == (ref["num_black_stones"], ref["num_white_stones"])
        assert m.side_to_move.get() == ref["side_to_move"]
        assert pipeline.to_SGF(m.full_board, m.side_to_move.get()) == ref["sgf"]               # to_SGF :781-810
        assert m.save_button.state == "active"                                                  # :575


@pytest.mark.parametrize("name", ["ex9.jpg", "no_circles.jpg", "ex7.jpg"])
def test_adapter_publishes_the_reference_globals(name):
    path = os.path.join(GOLDEN, "test_images", name)
    m = stand_in(path)
    state = gui_adapter.install(m)
    m.process_image()
    img = opipe.load_and_enhance(path)
    ref = opipe.process_image(img)
    ref["input"] = img
    check_against_oracle(m, ref)
    assert m.calls == ["draw_board", "draw_images", "draw_histogram"]                           # :576, :203, :204
    if ref["board_ready"]:
        # apply_black_thresh (:762-766): identify_board() only, on the cached detection
        m.black_stone_threshold = 250
        m.calls.clear()
        m.identify_board()
        ref2 = opipe.process_image(img, black_thr=250)
        np.testing.assert_array_equal(m.full_board, ref2["full_board"])
        assert pipeline.to_SGF(m.full_board, m.side_to_move.get()) == ref2["sgf"]
        assert m.calls == ["draw_histogram"]                                                    # :535
    else:
        assert any(s.startswith("Board not detected") for s in m.logged)
    state["det"].close()


def test_adapter_rotated_selection():
    """select_region (:677-723) + the rotate slider (:1077): rotate / crop / enhance run on the device inside the adapter."""
    path = os.path.join(GOLDEN, "test_images", "ex9.jpg")
    m = stand_in(path)
    state = gui_adapter.install(m)
    w, h = m.input_image_PIL.size
    m.selection_global = (4, 3, w - 6, h - 5)
    m.rotate_angle.set(1.5)
    m.threshold.set(pipeline.choose_threshold((w - 10, h - 8)))                # :721
    m.process_image()
    want = preprocess.enhance(preprocess.load_image(path), 70, 50, rotate_angle=1.5, selection=m.selection_global)
    ref = opipe.process_image(want, threshold=m.threshold.get())
    ref["input"] = want
    check_against_oracle(m, ref)
    state["det"].close()
