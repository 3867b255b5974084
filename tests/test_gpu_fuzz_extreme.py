"""Differential fuzzing at the EDGES of the parameter envelope i2s_detect_batch accepts (check_params in i2s_api.hip): accumulator
thresholds of a few votes (clouds of centre candidates), minDist below one pixel and beyond the image, one- and two-step radius
ranges, Hough-line thresholds of a few votes (hundreds of peaks), Canny thresholds at 0, equal, negative and beyond any gradient,
black thresholds at the ends, grid-spacing parameters that accept or reject everything.  Small images, so that the oracle stays quick;
everything the Detection carries must match bit for bit, and a capacity status must be one the oracle confirms."""
import os

import numpy as np
import pytest

import parity
from test_gpu_fuzz import _random_image
from img2sgf_amd.pipeline import Detector, Params
from oracle import cv_oracle as cvo
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("I2S_FUZZ_EXTREME_SEEDS", 40))


def _extreme_params(rng):
    lo = int(rng.choice([-5, 0, 1, 40, 255, 1020, 3000]))
    hi = lo + int(rng.choice([0, 1, 60, 2000]))
    rmax = int(rng.choice([1, 2, 3, 8, 30]))
    rmin = int(rng.integers(0, rmax))
    hc = (float(rng.choice([0.25, 1.0, 2.5, 10.0, 1e6])), int(rng.choice([1, 2, 3, 30, 100, 4000])), int(rng.choice([0, 1, 3, 8, 30, 200])), rmin, rmax)
    thr = int(rng.choice([1, 2, 5, 19, 74, 2000]))
    black = int(rng.choice([-1, 0, 1, 128, 254, 255, 300]))
    align = (2 + int(rng.integers(0, 2)), int(rng.integers(0, 2)))
    p = Params(canny_lo=lo, canny_hi=hi, hc_min_dist=hc[0], hc_param1=hc[1], hc_param2=hc[2], hc_min_radius=hc[3],
               hc_max_radius=hc[4], line_threshold=thr, black_threshold=black, alignment=align)
    return p, dict(canny=(lo, hi), hc=hc, threshold=thr, black_thr=black, alignment=align)


def run_extreme_seed(make_detector, seed, side=160, n_images=3):
    """One seed (also driven by the emulated twin in test_emu_pipeline.py, on smaller images)."""
    rng = np.random.default_rng(9000 + seed)
    imgs = [np.ascontiguousarray(_random_image(rng)[:side - 20, :side]) for _ in range(n_images)]
    params, okw = _extreme_params(rng)
    det = make_detector(n_images, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
    dets = det.detect_batch(imgs, params, full=True)
    over = [k for k, d in enumerate(dets) if d.status == 100]
    for k in over:
        ref = opipe.process_image(imgs[k], **okw)
        dbg = [cvo.hough_circles(b, *okw["hc"], debug=True)[1] for b in ref["blurs"]]
        assert (len(ref["circles_all"]) > 16384 or max(len(c) for c in ref["circles_per_variant"]) > 2048
                or max(len(d["est"]) for d in dbg) > 4096 or max(d["n_centers"] for d in dbg) > max(8192, det.max_w * det.max_h // 8)
                or len(ref["hlines"]) > 1024 or len(ref["vlines"]) > 1024), "capacity status without a capacity being exceeded"
    imgs = [im for k, im in enumerate(imgs) if k not in over]
    if imgs:
        parity.run_and_compare(det, imgs, params=params, oracle_kwargs=okw)
    det.close()


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_extreme_parameters(seed):
    run_extreme_seed(lambda nb, w, h: Detector(0, nb, w, h), seed)
