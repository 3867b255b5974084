"""Drop-in adapter for the reference's Tk application.

    import img2sgf                      # the reference module, unmodified
    import img2sgf_amd.gui_adapter as adapter
    adapter.install(img2sgf)            # process_image() / identify_board() now run on the GPU

`install(m)` replaces the two functions through which the GUI enters the hot path -- `process_image`
(img2sgf.py:117-204, called from 639, 723, 1077, 1134, 1140, 1178, 1181, 1191) and `identify_board` (497-543, called
from apply_black_thresh 765) -- with versions that run the detection through the C ABI and then assign exactly the
module globals the rest of the reference reads (draw_images 862-897, draw_board 900-952, draw_histogram 207-227,
cluster plotting 308-327, edit_board 955-1002, to_SGF 781-810).  The Pillow pre-processing inside process_image (rotate /
crop 110-114, contrast / brightness 141-149) runs on the device too, bit-exact with Pillow; everything else (Tk widgets,
logging, the board editor) keeps running the reference's own code.
"""
import types

import numpy as np

from . import pipeline, preprocess


def install(m, detector=None, lib=None, opencv=None):
    """Patch reference module `m` in place.  Returns the adapter's state (state["det"] = the Detector in use).

    opencv: the OpenCV release whose arithmetic the GPU path restates (Params.opencv_switches), or a dict of the three switches.
    None = what the cv2 the reference itself imported (`m.cv`) COMPUTES: the three switches are read off the live module by closed-form
    probes of the reference's own calls (pipeline.probe_cv2_switches) -- the patched application then answers as it did before the
    patch, whatever that module's version string says.  Where `m.cv` is not a module object with the three functions (a headless mock),
    or where the probe fails (an answer no known rule explains: Params.from_cv2 would raise), its `__version__` string decides
    (opencv_switches: release boundaries from memory) -- the failure is logged through `m.log` and kept in state["switches_from"] --
    and the package defaults where there is no version string either."""
    cvmod = getattr(m, "cv", None)
    how = "given"
    if isinstance(opencv, dict):
        switches = dict(opencv)
    elif opencv is not None:
        switches = pipeline.Params.opencv_switches(opencv)
    else:
        probe_error = None
        switches = None
        if isinstance(cvmod, types.ModuleType) and all(callable(getattr(cvmod, f, None)) for f in ("cvtColor", "GaussianBlur", "HoughLines")):
            try:
                switches, how = pipeline.probe_cv2_switches(cvmod), "probed"
            except Exception as e:       # an answer no known rule explains (I2sError) or a call the module refuses: say so, do not guess silently
                probe_error = "%s: %s" % (type(e).__name__, e)
        if switches is None:
            v = getattr(cvmod, "__version__", None)
            opencv = v if isinstance(v, str) and v[:1].isdigit() else None
            switches, how = (pipeline.Params.opencv_switches(opencv), "version string") if opencv else ({}, "package defaults")
            if probe_error:
                how += " (the probe of the live module failed: %s)" % probe_error
                warn = getattr(m, "log", None)
                if callable(warn):
                    warn("img2sgf_amd: could not read the OpenCV arithmetic off the installed cv2 (%s); using the %s" % (probe_error, how.split(" (")[0]))
    state = {"det": detector, "lib": lib, "opencv": opencv, "switches": switches, "switches_from": how}

    def _detector(w, h):
        d = state["det"]
        if d is None or d.max_w < w or d.max_h < h:
            if d is not None:
                d.close()
            d = state["det"] = pipeline.Detector(0, 1, max(w, 1024), max(h, 1024), lib=state["lib"])
        return d

    def _params():
        # cv.Canny(.., apertureSize=sobel.get(), L2gradient=(gradient.get()==2)) (img2sgf.py:164-165): the widgets behind the two are
        # hidden (:1142-1182), so every user runs 3 / L1, the one flavour the kernels implement -- anything else is refused, not ignored
        sobel, gradient = getattr(m, "sobel", None), getattr(m, "gradient", None)
        if sobel is not None and int(sobel.get()) != 3:
            raise pipeline.I2sError("Canny apertureSize = %d: the GPU path implements the 3 x 3 Sobel only (img2sgf.py:164)" % int(sobel.get()))
        if gradient is not None and int(gradient.get()) == 2:
            raise pipeline.I2sError("Canny L2gradient = True: the GPU path implements the L1 gradient norm only (img2sgf.py:165)")
        return pipeline.Params(
            canny_lo=int(m.edge_min.get()), canny_hi=int(m.edge_max.get()),      # img2sgf.py:163
            line_threshold=int(m.threshold.get()),                               # :259
            black_threshold=int(m.black_stone_threshold),                        # :515, 541
            alignment=(int(m.board_alignment[0]), int(m.board_alignment[1])),    # :543
            **switches)

    def _publish_board(det):
        m.detected_board = det.detected_board
        m.full_board = det.full_board
        m.stone_brightnesses = det.stone_brightnesses
        m.num_black_stones, m.num_white_stones = det.num_black_stones, det.num_white_stones
        m.side_to_move.set(det.side_to_move)                                     # :529-534

    def process_image():
        """img2sgf.py:117-204 with the OpenCV section (153-198) and find_grid() (546-576) on the GPU."""
        if not m.image_loaded:
            return
        p = _params()                    # first: settings the GPU path does not implement are refused before anything is touched
        m.found_grid = m.valid_grid = m.board_ready = False
        m.log("\nProcessing image")
        if m.rotate_angle.get() != 0:
            m.log("Rotated by " + str(m.rotate_angle.get()) + " degrees")
        m.log("Contrast = " + str(m.contrast.get()))
        m.log("Brightness = " + str(m.brightness.get()))
        from PIL import Image
        raw = np.array(m.input_image_PIL)                                                              # decoded source (:651)
        sel = getattr(m, "selection_global", None)
        xf = preprocess.xform(m.input_image_PIL.size, m.rotate_angle.get(), sel)                       # :110-114 on the device
        m.log("Rotating / cropping / enhancing / converting to greyscale / Canny / detecting circles / finding grid on the GPU")
        h, w = xf[1][3] - xf[1][1], xf[1][2] - xf[1][0]
        d = _detector(w, h)
        p.contrast, p.brightness = int(m.contrast.get()), int(m.brightness.get())                      # :141-149 on the device
        det = d.detect_batch([raw], p, full=True, xforms=[xf])[0]
        m.input_image_np = d.fetch_source(0, 1 if raw.ndim == 2 else 3)                                # :150
        m.region_PIL = Image.fromarray(m.input_image_np)
        m.grey_image_np = d.fetch_plane(0, "grey")                                                     # :153
        m.edge_detected_image_np = d.fetch_plane(0, "edges")                                           # :162
        m.edge_detected_image_PIL = Image.fromarray(m.edge_detected_image_np)                          # :166
        m.circles_removed_image_np = d.fetch_plane(0, "removed")                                       # :169-198
        m.circles_removed_image_PIL = Image.fromarray(m.circles_removed_image_np)                      # :200
        # find_grid() results (:546-576)
        m.hcentres, m.vcentres = det.hcentres, det.vcentres
        m.found_grid, m.valid_grid, m.board_ready = det.found_grid, det.valid_grid, det.board_ready
        m.circles = det.circles if len(det.circles) else []                      # filtered when the grid is valid (:554)
        m.vsize, m.hsize = det.vsize, det.hsize
        m.hcentres_complete, m.vcentres_complete = det.hcentres_complete, det.vcentres_complete
        m.hspace, m.vspace = det.hspace, det.vspace
        m.log("Found " + str(len(det.hlines)) + " distinct horizontal lines and " + str(len(det.vlines)) +
              " distinct vertical lines")                                         # :263
        if det.board_ready:
            _publish_board(det)
            m.save_button.configure(state=m.tk.ACTIVE)                           # :575
        else:
            m.log("Board not detected: " + det.status_text)
        m.draw_board()                                                           # :576
        m.draw_images()                                                          # :203
        m.draw_histogram(m.stone_brightnesses)                                   # :204

    def identify_board():
        """img2sgf.py:497-543 on the cached detection (apply_black_thresh, :762-766)."""
        p = _params()
        d = state["det"]
        det = d.classify(0, 1, p)[0]
        _publish_board(det)
        m.draw_histogram(m.stone_brightnesses)                                   # :535

    m.process_image = process_image
    m.identify_board = identify_board
    return state
