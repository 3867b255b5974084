"""tools/valu_model.py (profiles/r06_valu_model.md): the GPU-less instruction model still reproduces round 4's SQ_INSTS_VALU counters for
the product's sources -- a change of the compiled loops or of the replay that moves a kernel's figure shows up here, not in a ranking
nobody re-reads."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")


def test_model_matches_round4_counters():
    import valu_model as vm
    from img2sgf_amd import build
    rp = vm.replay([0])
    meas, _ = vm.measured()
    got = vm.model_all(vm.kernels_of(build.CSRC), rp)
    ratio = {k: r["valu"] / meas[k] for k, r in got.items()}
    for k, tol in (("k_sobel_nms_rows<0, true>", 0.05), ("k_sobel_nms_rows<2, true>", 0.05), ("k_edge_bins", 0.05), ("k_blur<true>", 0.07)):
        assert abs(ratio[k] - 1) <= tol, (k, ratio[k])
    assert 1.0 <= ratio["k_vote_centres<30>"] <= 1.15, ratio      # the unexplained +11 % of profiles/r06_valu_model.md: not to drift further
