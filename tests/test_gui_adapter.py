"""the reference's OWN application code running on top of the GPU path.  The reference module is imported
with cv2 / Tk mocked (only possible where /root/reference exists: the build container), img2sgf_amd.gui_adapter.install()
swaps its two hot-path entry points, and the reference's open_file() -> initialise_parameters() -> process_image() ->
to_SGF() chain must produce the oracle's SGF.  Uses the emulated build of the product sources (no GPU here)."""
import importlib.util
import os
import sys
from unittest.mock import MagicMock

import numpy as np
import pytest

import emu_util
from helpers import GOLDEN
from oracle import pipeline as opipe

REF = "/root/reference/img2sgf.py"
pytestmark = pytest.mark.skipif(not os.path.exists(REF), reason="reference checkout not present (GPU box)")


class Var:
    def __init__(self, v):
        self.v = v

    def get(self):
        return self.v

    def set(self, v):
        self.v = v


def load_reference():
    for name in ["cv2", "tkinter", "tkinter.messagebox", "tkinter.filedialog", "tkinter.scrolledtext",
                 "matplotlib.backends.backend_tkagg", "pyscreenshot", "PIL.ImageTk", "PIL.ImageGrab"]:
        sys.modules[name] = MagicMock()
    argv, sys.argv = sys.argv, ["img2sgf.py"]
    sys.dont_write_bytecode = True
    try:
        spec = importlib.util.spec_from_file_location("ref_img2sgf_gui", REF)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        sys.argv = argv
    m.log = lambda *a, **k: None
    m.draw_board = m.draw_images = m.draw_histogram = lambda *a, **k: None
    m.rotate_angle, m.contrast, m.brightness = Var(0), Var(70), Var(50)
    m.threshold, m.edge_min, m.edge_max, m.sobel, m.gradient = Var(80), Var(50), Var(200), Var(3), Var(1)
    m.side_to_move = Var(1)
    return m


@pytest.mark.parametrize("name", ["ex9.jpg", "no_circles.jpg"])
def test_reference_app_on_gpu_path(name):
    from img2sgf_amd import gui_adapter
    m = load_reference()
    gui_adapter.install(m, lib=emu_util.emu_library())
    path = os.path.join(GOLDEN, "test_images", name)
    m.open_file(path)                                  # reference code: Image.open, initialise_parameters, process_image
    ref = opipe.process_image(opipe.load_and_enhance(path))
    assert m.threshold.get() == ref["threshold"]
    assert bool(m.board_ready) == ref["board_ready"]
    np.testing.assert_array_equal(m.edge_detected_image_np, ref["edges"])
    np.testing.assert_array_equal(m.circles_removed_image_np, ref["circles_removed"])
    if ref["board_ready"]:
        assert m.to_SGF(m.full_board) == ref["sgf"]    # the reference's own writer on our matrix
        # apply_black_thresh path: identify_board only
        m.black_stone_threshold = 250
        m.identify_board()
        ref2 = opipe.process_image(opipe.load_and_enhance(path), black_thr=250)
        assert m.to_SGF(m.full_board) == ref2["sgf"]


def test_reference_app_rotated_selection_on_gpu_path():
    """The rotate slider and a region selection: crop_and_rotate_image runs on the device inside the adapter and must leave
    the reference's globals exactly as its own Pillow code would."""
    from img2sgf_amd import gui_adapter, preprocess
    m = load_reference()
    gui_adapter.install(m, lib=emu_util.emu_library())
    path = os.path.join(GOLDEN, "test_images", "ex9.jpg")
    m.open_file(path)
    w, h = m.input_image_PIL.size
    m.selection_global = (4, 3, w - 6, h - 5)
    m.rotate_angle.set(1.5)
    m.process_image()
    want = preprocess.enhance(preprocess.load_image(path), 70, 50, rotate_angle=1.5, selection=m.selection_global)
    np.testing.assert_array_equal(m.input_image_np, want)
    np.testing.assert_array_equal(np.array(m.region_PIL), want)
    ref = opipe.process_image(want, threshold=m.threshold.get())
    assert bool(m.board_ready) == ref["board_ready"]
    np.testing.assert_array_equal(m.edge_detected_image_np, ref["edges"])
    if ref["board_ready"]:
        assert m.to_SGF(m.full_board) == ref["sgf"]
