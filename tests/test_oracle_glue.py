"""Pins oracle/glue.py (the restatement of img2sgf.py:268-576, 606-613, 781-810) against vectors
produced by the reference itself (tests/golden/make_glue_golden.py)."""
import numpy as np
import pytest

from oracle import glue
from helpers import load_glue_golden, synth_grey

G = load_glue_golden()


def run_oracle_glue(case):
    grey = synth_grey(case["w"], case["h"], case["seed"])
    hl = np.array(case["hlines"], np.float32).reshape(-1, 1)
    vl = np.array(case["vlines"], np.float32).reshape(-1, 1)
    circles = [np.array(c, np.float32) for c in case["circles"]]
    hc = glue.cluster_centres(hl)
    vc = glue.cluster_centres(vl)
    out = dict(found_grid=len(hc) > 0 and len(vc) > 0, hcentres=list(hc), vcentres=list(vc))
    g = glue.validate_grid(hc, vc, circles)
    out.update(valid_grid=g["valid"], hsize=g["hsize"], vsize=g["vsize"], board_ready=False)
    if g["valid"]:
        out.update(hcentres_complete=list(g["hc"]), vcentres_complete=list(g["vc"]),
                   hspace=g["hspace"], vspace=g["vspace"],
                   kept_circles=[[float(v) for v in c] for c in g["circles"]])
        if g["hsize"] <= 19 and g["vsize"] <= 19:
            ib = glue.identify_board(grey, g, case["black_thr"], case["alignment"])
            out.update(board_ready=True, detected_board=ib["detected_board"].astype(int).tolist(),
                       full_board=ib["full_board"].astype(int).tolist(),
                       stone_brightnesses=list(ib["stone_brightnesses"]),
                       num_black_stones=ib["num_black_stones"], num_white_stones=ib["num_white_stones"],
                       side_to_move=ib["side_to_move"], sgf=glue.to_sgf(ib["full_board"], ib["side_to_move"]))
    return out


@pytest.mark.parametrize("entry", G["cases"], ids=[e["case"]["name"] for e in G["cases"]])
def test_glue_case(entry):
    got = run_oracle_glue(entry["case"])
    exp = entry["expect"]
    for k, v in exp.items():
        if isinstance(v, float) or (isinstance(v, list) and v and isinstance(v[0], float)):
            np.testing.assert_array_equal(np.asarray(got[k], np.float64), np.asarray(v, np.float64), err_msg=k)
        else:
            assert got[k] == v, k


def test_choose_threshold():
    for w, h, t in G["helpers"]["choose_threshold"]:
        assert glue.choose_threshold(w, h) == t


def test_closest_index():
    for a, xs, i in G["helpers"]["closest_index"]:
        assert glue.closest_index(a, xs) == i


def test_to_sgf():
    for side, board, s in G["helpers"]["to_SGF"]:
        assert glue.to_sgf(np.array(board), side) == s
