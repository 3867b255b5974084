#!/usr/bin/env python3
"""Host-path rate on real scans (the 18 reference fixtures x 8, ragged sizes 110x102 .. 1265x1245, RGB, host numpy inputs):
shuffled input order, with and without the area-sorted pass formation (Params.schedule)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img2sgf_amd import preprocess                       # noqa: E402
from img2sgf_amd.pipeline import Detector, Params        # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden", "test_images")
names = ["ex%d.jpg" % i for i in range(1, 18)] + ["no_circles.jpg"]
raws = [preprocess.enhance(preprocess.load_image(os.path.join(GOLDEN, n))) for n in names]
imgs = raws * 8
perm = np.random.default_rng(0).permutation(len(imgs))
mixed = [imgs[i] for i in perm]
MB = int(os.environ.get("I2S_RAGGED_PASS", 16))
det = Detector(0, MB, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
for name, params in (("input order", Params()), ("scheduled", Params(schedule=True))) * 2:
    det.detect_batch(mixed[:MB], params, full=False)
    t = time.perf_counter()
    det.detect_batch(mixed, params, full=False)
    dt = time.perf_counter() - t
    print("%-12s %.1f ms for %d images (%.0f img/s, %.0f Mpx/s)" % (name, dt * 1e3, len(mixed), len(mixed) / dt,
                                                                  sum(i.shape[0] * i.shape[1] for i in mixed) / dt / 1e6))
det.close()
from img2sgf_amd.pipeline import StreamedDetector      # noqa: E402
ref = None
for n_streams in (2, 3, 4):
    sd = StreamedDetector(0, n_streams, MB, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
    sd.detect_batch(mixed[:3 * MB])
    for rep in range(2):
        t = time.perf_counter()
        boards = sd.detect_batch(mixed)
        dt = time.perf_counter() - t
    print("%d streams    %.1f ms for %d images (%.0f img/s, %.0f Mpx/s)" % (n_streams, dt * 1e3, len(mixed), len(mixed) / dt,
                                                                        sum(i.shape[0] * i.shape[1] for i in mixed) / dt / 1e6))
    sd.close()
