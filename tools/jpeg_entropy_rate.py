#!/usr/bin/env python3
"""File bytes -> board records: Huffman decoding on the device (one lane per file) against host threads, by pass size.
The 18 reference fixtures x 16 (progressive ones included) and Pillow-encoded 1024x1024 diagrams."""
import io
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img2sgf_amd import pipeline, synth                       # noqa: E402
from img2sgf_amd.pipeline import Detector, Params             # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "test_images")
fixtures = []
for n in sorted(os.listdir(G)):
    with open(os.path.join(G, n), "rb") as f:
        fixtures.append(f.read())
fixtures = fixtures * 16
diagrams = []
for s in range(32):
    b = io.BytesIO()
    Image.fromarray(synth.synth_diagram(s)[0]).save(b, "JPEG", quality=90)
    diagrams.append(b.getvalue())
diagrams = diagrams * 8


def rate(blobs, mb, host, size):
    det = Detector(0, mb, size, size)
    p = Params(schedule=True, jpeg_entropy_device=not host)
    det.detect_jpeg(blobs[:mb], p, full=False)
    t = time.perf_counter()
    out = det.detect_jpeg(blobs, p, full=False)
    dt = time.perf_counter() - t
    det.close()
    return len(blobs) / dt, out


for name, blobs, size in (("18 fixtures x 16", fixtures, 1300), ("1024x1024 diagrams, q90", diagrams, 1024)):
    print("%s: %d files, %.1f MB" % (name, len(blobs), sum(len(b) for b in blobs) / 1e6))
    ref = None
    for mb in (16, 64, 256):
        rh, oh = rate(blobs, mb, True, size)
        rd, od = rate(blobs, mb, False, size)
        assert all(bytes(x) == bytes(y) for x, y in zip(oh, od))
        print("  pass of %3d files: host threads %6.0f files/s   device lanes %6.0f files/s" % (mb, rh, rd))
