"""Host pre-processing (Pillow) matches the oracle's copy of the reference recipe; SHA-256 of the enhanced arrays pins
the Pillow behaviour the parity fixtures were generated with."""
import hashlib
import json
import os

import numpy as np

from helpers import GOLDEN
from img2sgf_amd import preprocess
from oracle import pipeline as opipe

NAMES = ["ex1.jpg", "ex7.jpg", "ex9.jpg", "no_circles.jpg"]


def test_matches_oracle_recipe():
    for n in NAMES:
        p = os.path.join(GOLDEN, "test_images", n)
        np.testing.assert_array_equal(preprocess.load_and_enhance(p), opipe.load_and_enhance(p))


def test_enhanced_hashes():
    with open(os.path.join(GOLDEN, "enhanced_sha256.json")) as f:
        want = json.load(f)
    for n, h in want.items():
        a = preprocess.load_and_enhance(os.path.join(GOLDEN, "test_images", n))
        assert hashlib.sha256(a.tobytes()).hexdigest() == h, n
