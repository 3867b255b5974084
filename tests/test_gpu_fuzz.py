"""Differential fuzzing of the HIP path against the oracle: random image kinds (noise, smooth shapes, synthetic diagrams,
crops of the reference's scans), ragged sizes from 1x1 up, mixed grey / colour batches and random NON-default parameters
(Canny thresholds, HoughCircles arguments, Hough-lines threshold, black threshold, alignment).  Everything the Detection
carries and every plane of the last device pass must match bit for bit."""
import os

import numpy as np
import pytest

import parity
import switches
from helpers import GOLDEN
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu

_SCANS = {}


def _scan(name):
    if name not in _SCANS:
        _SCANS[name] = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", name))
    return _SCANS[name]


def _random_image(rng):
    kind = rng.integers(0, 6)
    h, w = int(rng.integers(1, 330)), int(rng.integers(1, 330))
    if kind == 0:                                       # uniform noise, sometimes tiny
        if rng.random() < 0.3:
            h, w = int(rng.integers(1, 12)), int(rng.integers(1, 12))
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    elif kind == 1:                                     # smooth ramps + discs + bars
        yy, xx = np.mgrid[0:h, 0:w]
        img = ((xx * rng.uniform(0, 1.5) + yy * rng.uniform(0, 1.5)) % 256).astype(np.float64)
        for _ in range(int(rng.integers(1, 8))):
            cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(3, 30)
            img[(xx - cx) ** 2 + (yy - cy) ** 2 <= r * r] = rng.integers(0, 256)
        for _ in range(int(rng.integers(0, 6))):
            if rng.random() < 0.5:
                img[:, int(rng.integers(0, w)):][:, :2] = rng.integers(0, 256)
            else:
                img[int(rng.integers(0, h)):][:2, :] = rng.integers(0, 256)
        img = np.clip(img, 0, 255).astype(np.uint8)
    elif kind in (2, 3):                                # synthetic diagram, cropped / padded, maybe noisy
        base = synth.synth_diagram(int(rng.integers(0, 1000)), noisy=kind == 3, geom=synth.GEOM_SMALL)[0]
        y0, x0 = int(rng.integers(0, 40)), int(rng.integers(0, 40))
        img = np.ascontiguousarray(base[y0:y0 + max(h, 60), x0:x0 + max(w, 60)])
    else:                                               # window of a reference scan (RGB)
        src = _scan("ex%d.jpg" % int(rng.integers(1, 18)))
        sh, sw = src.shape[:2]
        hh, ww = min(h + 40, sh), min(w + 40, sw)
        y0, x0 = int(rng.integers(0, sh - hh + 1)), int(rng.integers(0, sw - ww + 1))
        img = np.ascontiguousarray(src[y0:y0 + hh, x0:x0 + ww])
    if img.ndim == 2 and rng.random() < 0.25:           # grey content as a 3-channel image with unequal channels
        img = np.ascontiguousarray(np.stack([img, img[::-1], np.roll(img, 3, axis=1)], axis=-1))
    return img


def _random_params(rng, seed=0):
    if seed % 3 == 2:
        # every third seed: a non-default OpenCV-version switch set, with default or random other parameters
        sw = switches.NAMES[(seed // 3) % len(switches.NAMES)]
        if rng.random() < 0.5:
            return switches.params(sw), dict(compat=switches.compat(sw))
        p, okw = _random_params(rng, 0)
        for k, v in switches.params_kwargs(sw).items():
            setattr(p, k, v)
        return p, dict(okw, compat=switches.compat(sw))
    if rng.random() < 0.35:
        return Params(), {}
    lo = int(rng.integers(5, 120))
    hi = lo + int(rng.integers(0, 200))
    p1 = int(rng.integers(20, 200))
    hc = (float(rng.choice([4.0, 10.0, 17.5])), p1, int(rng.integers(15, 60)), int(rng.integers(0, 6)), int(rng.integers(8, 31)))
    thr = int(rng.integers(20, 140))
    black = int(rng.integers(40, 220))
    align = (2 + int(rng.integers(0, 2)), int(rng.integers(0, 2)))      # (LEFT | RIGHT, TOP | BOTTOM), img2sgf.py:86-87
    p = Params(canny_lo=lo, canny_hi=hi, hc_min_dist=hc[0], hc_param1=hc[1], hc_param2=hc[2], hc_min_radius=hc[3],
               hc_max_radius=hc[4], line_threshold=thr, black_threshold=black, alignment=align)
    return p, dict(canny=(lo, hi), hc=hc, threshold=thr, black_thr=black, alignment=align)


N_SEEDS = int(os.environ.get("I2S_FUZZ_SEEDS", 60))       # raise for a longer hunt


def _gpu_detector(nb, w, h):
    return Detector(0, nb, w, h)


def run_fuzz_seed(make_detector, seed):
    """One seed (also driven on the emulated kernels: tests/test_emu_pipeline.py, tests/stress/emulated_fuzz.py)."""
    rng = np.random.default_rng(1000 + seed)
    imgs = [_random_image(rng) for _ in range(4)]
    params, okw = _random_params(rng, seed)
    det = make_detector(4, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
    dets = det.detect_batch(imgs, params, full=True)
    over = [k for k, d in enumerate(dets) if d.status == 100]
    if over:
        # capacities (include/i2s.h: I2S_ST_CAPACITY) are reported, never silently truncated: the oracle must agree that one of
        # them was exceeded (16384 circles per image; per HoughCircles call 2048 circles and 4096 supported estimates per
        # started megapixel of the context -- these contexts are below one megapixel -- and max(8192, area / 8) accumulator maxima)
        from oracle import cv_oracle as cvo
        for k in over:
            ref = opipe.process_image(imgs[k], **okw)
            worst = max(len(c) for c in ref["circles_per_variant"])
            dbg = [cvo.hough_circles(b, *okw.get("hc", (10, 100, 30, 1, 30)), debug=True)[1] for b in ref["blurs"]]
            assert (len(ref["circles_all"]) > 16384 or worst > 2048 or max(len(d["est"]) for d in dbg) > 4096
                    or max(d["n_centers"] for d in dbg) > max(8192, det.max_w * det.max_h // 8)), "capacity status without a capacity being exceeded"
        imgs = [im for k, im in enumerate(imgs) if k not in over]
    if imgs:
        parity.run_and_compare(det, imgs, params=params, internals=set(okw) <= {"compat"}, oracle_kwargs=okw)
    det.close()


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_against_oracle(seed):
    run_fuzz_seed(_gpu_detector, seed)


def run_preprocessing_fuzz_seed(make_detector, seed):
    """Random rotate / crop / contrast / brightness on the device against Pillow itself (staged source bit for bit), then the
    detection against the oracle run on Pillow's result."""
    from PIL import Image
    from img2sgf_amd import preprocess
    rng = np.random.default_rng(5000 + seed)
    imgs, xfs, wants = [], [], []
    contrast, brightness = int(rng.integers(0, 101)), int(rng.integers(0, 101))
    for _ in range(3):
        img = _random_image(rng)
        h, w = img.shape[:2]
        angle = float(rng.choice([0.0, 90.0, 180.0, -90.0])) if rng.random() < 0.3 else float(rng.uniform(-180, 180))
        if rng.random() < 0.4:
            sel = None
        else:
            x1, y1 = int(rng.integers(-5, max(w // 2, 1))), int(rng.integers(-5, max(h // 2, 1)))
            sel = (x1, y1, x1 + int(rng.integers(1, w + 8)), y1 + int(rng.integers(1, h + 8)))
        imgs.append(img)
        xfs.append(preprocess.xform((w, h), angle, sel))
        wants.append(preprocess.enhance(Image.fromarray(img), contrast, brightness, rotate_angle=angle, selection=sel))
    det = make_detector(3, max(x[1][2] - x[1][0] for x in xfs), max(x[1][3] - x[1][1] for x in xfs))
    dets = det.detect_batch(imgs, Params(contrast=contrast, brightness=brightness), xforms=xfs)
    for k, (img, want, d) in enumerate(zip(imgs, wants, dets)):
        np.testing.assert_array_equal(det.fetch_source(k, 1 if img.ndim == 2 else 3), want, err_msg="image %d" % k)
        if d.status != 100:
            parity.compare_detection(d, opipe.process_image(want))
    det.close()


@pytest.mark.parametrize("seed", range(max(N_SEEDS // 3, 1)))
def test_fuzz_device_preprocessing_against_pillow(seed):
    run_preprocessing_fuzz_seed(_gpu_detector, seed)
