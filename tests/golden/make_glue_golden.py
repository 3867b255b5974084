#!/usr/bin/env python3
"""Generates tests/golden/glue_golden.json by running the REFERENCE's own glue code.

Runs only in the build container (needs /root/reference); the output JSON is data
(inputs + the reference's outputs) and is what travels.  The reference module is loaded
with cv2 / tkinter / matplotlib-Tk / pyscreenshot replaced by MagicMock: the
OpenCV rows are NOT exercised here (cv2 is absent) -- `find_lines` is replaced by a
function returning the rho arrays given in the case, everything downstream
(find_grid -> cluster_lines (sklearn) -> validate_grid -> identify_board -> to_SGF,
img2sgf.py:258-576, 781-810) runs for real.

Usage:  python tests/golden/make_glue_golden.py
"""
import importlib.util
import json
import os
import sys
from unittest.mock import MagicMock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/img2sgf.py"


def load_reference():
    for name in ["cv2", "tkinter", "tkinter.messagebox", "tkinter.filedialog", "tkinter.scrolledtext",
                 "matplotlib.backends.backend_tkagg", "pyscreenshot", "PIL.ImageTk", "PIL.ImageGrab"]:
        sys.modules[name] = MagicMock()
    sys.argv = ["img2sgf.py"]
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location("ref_img2sgf", REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.log = lambda *a, **k: None
    m.draw_board = lambda *a, **k: None
    m.draw_histogram = lambda *a, **k: None
    return m


def synth_grey(w, h, seed):
    """Deterministic integer-only texture; tests regenerate it from (w,h,seed)."""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    a, b, c = 7 + seed % 13, 11 + seed % 17, 23 + seed % 29
    v = (x * a + y * b + ((x * y) % c) * 5 + ((x // 9 + y // 7 + seed) % 2) * 120) % 256
    return v.astype(np.uint8)


def run_case(m, case):
    w, h, seed = case["w"], case["h"], case["seed"]
    grey = synth_grey(w, h, seed)
    state = {"side": 0}
    m.threshold.get = lambda: case["threshold"]
    m.side_to_move.get = lambda: state["side"]
    m.side_to_move.set = lambda v: state.__setitem__("side", int(v))
    hl = np.array(case["hlines"], np.float32).reshape(-1, 1)
    vl = np.array(case["vlines"], np.float32).reshape(-1, 1)

    def fake_find_lines(threshold, direction):
        a = hl if direction == m.Direction.H else vl
        return [] if len(a) == 0 else a.copy()

    m.find_lines = fake_find_lines
    m.grey_image_np = grey
    m.circles = np.array(case["circles"], np.float32).reshape(-1, 3)
    if len(m.circles) == 0:
        m.circles = []
    m.black_stone_threshold = case["black_thr"]
    m.board_alignment = [m.Alignment(case["alignment"][0]), m.Alignment(case["alignment"][1])]
    m.found_grid = False
    m.valid_grid = False
    m.board_ready = False
    m.find_grid()
    out = dict(found_grid=bool(m.found_grid), valid_grid=bool(m.valid_grid), board_ready=bool(m.board_ready),
               hcentres=[float(v) for v in m.hcentres], vcentres=[float(v) for v in m.vcentres],
               hsize=int(m.hsize), vsize=int(m.vsize))
    if m.valid_grid:
        out.update(hcentres_complete=[float(v) for v in m.hcentres_complete],
                   vcentres_complete=[float(v) for v in m.vcentres_complete],
                   hspace=float(m.hspace), vspace=float(m.vspace),
                   kept_circles=[[float(v) for v in c] for c in m.circles])
    if m.board_ready:
        out.update(detected_board=np.asarray(m.detected_board).astype(int).tolist(),
                   full_board=np.asarray(m.full_board).astype(int).tolist(),
                   stone_brightnesses=[float(v) for v in m.stone_brightnesses],
                   num_black_stones=int(m.num_black_stones), num_white_stones=int(m.num_white_stones),
                   side_to_move=int(state["side"]), sgf=m.to_SGF(m.full_board))
    return out


def jitter_lines(rng, centres, per=(1, 4), spread=3):
    out = []
    for c in centres:
        for _ in range(int(rng.integers(per[0], per[1] + 1))):
            out.append(float(int(c) + int(rng.integers(-spread, spread + 1))))
    return out


def make_cases():
    rng = np.random.default_rng(20260928)
    cases = []

    def add(name, w, h, hcs, vcs, ncirc, **kw):
        seed = len(cases)
        hl = kw.pop("hl", None)
        vl = kw.pop("vl", None)
        if hl is None:
            hl = jitter_lines(rng, hcs)
        if vl is None:
            vl = jitter_lines(rng, vcs)
        circ = kw.pop("circles", None)
        if circ is None:
            circ = []
            for _ in range(ncirc):
                # circle centres are always k+0.5 in the reference's data (HoughCircles output)
                cx = float(int(rng.integers(-5, w + 5))) + 0.5
                cy = float(int(rng.integers(-5, h + 5))) + 0.5
                r = float(np.float32(rng.integers(10, 300) / 10.0))
                circ.append([cx, cy, r])
        cases.append(dict(name=name, w=w, h=h, seed=seed, threshold=int(kw.pop("threshold", 50)),
                          black_thr=kw.pop("black_thr", 128), alignment=kw.pop("alignment", [2, 0]),
                          hlines=hl, vlines=vl, circles=circ))

    full = [20 + 24 * k for k in range(19)]
    add("full19", 480, 480, full, full, 60)
    add("full19_b", 480, 480, full, full, 200, black_thr=100)
    add("full19_exactlines", 480, 480, None, None, 40, hl=[float(v) for v in full], vl=[float(v) for v in full])
    # gaps to fill
    g1 = [v for i, v in enumerate(full) if i not in (3, 4, 10)]
    g2 = [v for i, v in enumerate(full) if i not in (0, 7, 8, 9, 17)]
    add("gaps", 480, 480, g1, g2, 80)
    add("gaps2", 480, 480, g2, g1, 80, alignment=[3, 1])
    # 21 / 20 lines -> truncate
    l21 = [8 + 22 * k for k in range(21)]
    l20 = [8 + 22 * k for k in range(20)]
    add("trunc21", 470, 470, l21, l21, 50)
    add("trunc20", 470, 470, l20, l21, 50)
    add("trunc20v", 470, 470, l21, l20, 50)
    l22 = [8 + 21 * k for k in range(22)]
    add("too_many_22", 480, 480, l22, full, 30)
    add("too_many_22v", 480, 480, full, l22, 30)
    # part boards
    part9 = [20 + 24 * k for k in range(9)]
    add("part_9x19", 480, 240, part9, full, 40)
    add("part_9x19_bottom", 480, 240, part9, full, 40, alignment=[2, 1])
    add("part_19x7_right", 200, 480, full, part9[:7], 40, alignment=[3, 0])
    add("part_5x5_rb", 150, 150, part9[:5], part9[:5], 20, alignment=[3, 1])
    # degenerate
    add("no_lines", 100, 100, [], [], 5)
    add("one_h_line", 100, 100, None, None, 5, hl=[50.0], vl=[10.0, 30.0, 50.0])
    add("one_cluster_each", 100, 100, None, None, 5, hl=[50.0, 51.0], vl=[10.0, 12.0])
    add("h_only", 100, 100, None, None, 5, hl=[10.0, 30.0, 50.0], vl=[])
    add("too_close", 200, 200, None, None, 5, hl=[10.0, 21.0, 50.0, 90.0], vl=[10.0, 20.5, 31.0, 45.0],
        )
    add("spacing_lt10", 200, 200, None, None, 5, hl=[10.0, 15.0, 26.0, 27.0, 60.0], vl=[10.0, 40.0, 70.0])
    add("no_circles", 480, 480, full, full, 0)
    add("weird_gaps_gt21", 900, 480, [10 + 12 * k for k in range(3)] + [400.0, 880.0], full, 10)
    # randomised: random line sets with merges, gaps; random circles incl. outside / wrong radius
    for t in range(60):
        n_h = int(rng.integers(2, 23))
        n_v = int(rng.integers(2, 23))
        sp_h = int(rng.integers(11, 40))
        sp_v = int(rng.integers(11, 40))
        hcs = [15 + sp_h * k for k in range(n_h)]
        vcs = [15 + sp_v * k for k in range(n_v)]
        # drop some lines
        hcs = [c for c in hcs if rng.random() > 0.15] or hcs[:2]
        vcs = [c for c in vcs if rng.random() > 0.15] or vcs[:2]
        w = 30 + sp_v * n_v
        h = 30 + sp_h * n_h
        add("rand%02d" % t, w, h, hcs, vcs, int(rng.integers(0, 120)),
            black_thr=int(rng.integers(60, 200)),
            alignment=[int(rng.integers(2, 4)), int(rng.integers(0, 2))])
    return cases


def main():
    m = load_reference()
    cases = make_cases()
    out = []
    for c in cases:
        exp = run_case(m, c)
        out.append(dict(case=c, expect=exp))
    # helper-level vectors
    helpers = dict(
        choose_threshold=[[w, h, m.choose_threshold(MagicMock(size=(w, h)))]
                          for (w, h) in [(750, 747), (239, 175), (110, 102), (1265, 1245), (1024, 1024),
                                         (10, 10), (5000, 4000), (2355, 2355), (51, 52), (52, 51)]],
        closest_index=[[a, xs, int(m.closest_index(a, xs))]
                       for a, xs in [(5.0, [1.0, 4.0, 6.0, 9.0]), (0.0, [1.0, 4.0]), (10.0, [1.0, 4.0]),
                                     (2.5, [1.0, 4.0]), (4.0, [1.0, 4.0, 7.0]), (5.5, [1.0, 4.0, 7.0]),
                                     (1.0, [1.0]), (6.5, [1.0, 4.0, 9.0])]],
    )
    state = {"side": 1}
    m.side_to_move.get = lambda: state["side"]
    sg = []
    for side in (1, 2):
        state["side"] = side
        b = np.zeros((19, 19))
        sg.append([side, b.astype(int).tolist(), m.to_SGF(b)])
        b[3, 3] = 1
        b[15, 3] = 2
        b[18, 18] = 2
        sg.append([side, b.astype(int).tolist(), m.to_SGF(b)])
        b2 = np.zeros((19, 19))
        b2[0, 0] = 2
        sg.append([side, b2.astype(int).tolist(), m.to_SGF(b2)])
    helpers["to_SGF"] = sg
    path = os.path.join(HERE, "glue_golden.json")
    with open(path, "w") as f:
        json.dump(dict(cases=out, helpers=helpers), f)
    nready = sum(1 for o in out if o["expect"]["board_ready"])
    nvalid = sum(1 for o in out if o["expect"]["valid_grid"])
    print("wrote", path, len(out), "cases;", nvalid, "valid grids;", nready, "boards")


if __name__ == "__main__":
    main()
