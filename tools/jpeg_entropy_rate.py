#!/usr/bin/env python3
"""File bytes -> board records (i2s_detect_jpeg_batch end to end) by where the Huffman decoding runs
(i2s_params.jpeg_entropy_device: 0 host threads, 1 sequential files in parallel on the device, 2 = 1 + progressive files one
lane each), by pass size.  The 18 reference fixtures x 16 (four progressive ones among them), Pillow-encoded 1024x1024 diagrams
and a blank page (the worst case for the parallel decoder's iteration count).  --lanes adds mode 2 (slow); --device-only runs
mode 1 at 256 files per pass alone (for a kernel trace)."""
import io
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img2sgf_amd import synth                                 # noqa: E402
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
b = io.BytesIO()
Image.fromarray(np.full((1024, 1024, 3), 255, np.uint8)).save(b, "JPEG", quality=90)
blank = [b.getvalue()] * 64

MODES = [(0, "host threads"), (1, "device")] + ([(2, "device + lanes")] if "--lanes" in sys.argv else [])
if "--device-only" in sys.argv:                # for rocprofv3: the default path alone
    MODES = [(1, "device")]
PASSES = (256,) if "--device-only" in sys.argv else (16, 64, 256)


def rate(blobs, mb, mode, size):
    det = Detector(0, mb, size, size)
    p = Params(schedule=True, jpeg_entropy_device=mode)
    det.detect_jpeg(blobs[:mb], p, full=False)
    t = time.perf_counter()
    out = det.detect_jpeg(blobs, p, full=False)
    dt = time.perf_counter() - t
    rounds, ms = det.jpeg_last_rounds(), det.jpeg_last_timing()
    det.close()
    return len(blobs) / dt, out, rounds, ms


for name, blobs, size in (("18 fixtures x 16", fixtures, 1300), ("1024x1024 diagrams, q90", diagrams, 1024), ("blank 1024x1024 page", blank, 1024)):
    print("%s: %d files, %.1f MB" % (name, len(blobs), sum(len(b) for b in blobs) / 1e6))
    for mb in PASSES:
        if mb > len(blobs):
            continue
        line, ref = "  pass of %3d files:" % mb, None
        for mode, label in MODES:
            r, out, rounds, ms = rate(blobs, mb, mode, size)
            if ref is None:
                ref = out
            assert all(bytes(x) == bytes(y) for x, y in zip(ref, out))
            line += "   %s %6.0f files/s" % (label, r) + (" (%d rounds)" % rounds if mode else "")
            line += " [parse %.1f, entropy host %.1f, entropy wait %.1f, call %.1f ms]" % tuple(ms)
        print(line)
