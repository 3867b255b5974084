"""Differential fuzzing at medium sizes (500 .. 1700 pixels a side): the small-image fuzz rarely leaves one accumulator tile, one
hysteresis tile row or one blur band; here the same random image kinds are blown up by whole factors (blocks: thick strokes, circles
of every radius up to the limit), tiled into mosaics and sprinkled with noise, so that every kernel's tile seams, the multi-tile
vote / radius / erase paths and the worklists see irregular content.  Every plane, accumulator, list and record against the oracle."""
import os

import numpy as np
import pytest

import parity
from test_gpu_fuzz import _random_image, _random_params
from img2sgf_amd.pipeline import Detector

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("I2S_FUZZ_MEDIUM_SEEDS", 6))


def _medium_image(rng):
    im = _random_image(rng)
    k = int(rng.integers(2, 6))
    im = np.kron(im, np.ones((k, k) + (1,) * (im.ndim - 2), np.uint8))
    if rng.random() < 0.4:                                   # a mosaic of itself, mirrored
        im = np.concatenate([im, im[:, ::-1]], axis=1) if rng.random() < 0.5 else np.concatenate([im, im[::-1]], axis=0)
    im = im[:int(rng.integers(500, 1700)), :int(rng.integers(500, 1700))]
    r = rng.random()
    if r < 0.3:                                              # salt and pepper on a fraction of the pixels
        m = rng.random(im.shape[:2]) < rng.choice([0.001, 0.01, 0.05])
        im = im.copy()
        im[m] = rng.integers(0, 256, im[m].shape, dtype=np.uint8)
    elif r < 0.5:                                            # Gaussian noise
        im = np.clip(im.astype(np.float32) + rng.normal(0, rng.choice([2, 6, 20]), im.shape), 0, 255).astype(np.uint8)
    return np.ascontiguousarray(im)


def run_medium_seed(make_detector, seed, n_images=2):
    """One seed (also driven on the emulated kernels by tests/stress/emulated_fuzz.py)."""
    rng = np.random.default_rng(130000 + seed)
    imgs = [_medium_image(rng) for _ in range(n_images)]
    params, okw = _random_params(rng, seed)
    det = make_detector(n_images, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
    boards = det.detect_batch(imgs, params, full=False)
    imgs = [im for k, im in enumerate(imgs) if boards[k].status != 100]
    if imgs:
        parity.run_and_compare(det, imgs, params=params, internals=set(okw) <= {"compat"}, oracle_kwargs=okw)
    det.close()


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_medium_sizes(seed):
    run_medium_seed(lambda nb, w, h: Detector(0, nb, w, h), seed)
