#!/usr/bin/env python3
"""End-to-end rate from JPEG file bytes (the reference's 18 JPEG fixtures x 8): Pillow decode + host arrays versus
i2s_detect_jpeg_batch (Huffman on the host, the rest of the decoder and the detection on the GPU)."""
import io
import os
import sys
import time

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img2sgf_amd import pipeline                       # noqa: E402
from img2sgf_amd.pipeline import Detector, Params      # noqa: E402

G = os.path.join(ROOT, "tests", "golden", "test_images")
blobs = []
for n in sorted(os.listdir(G)):
    with open(os.path.join(G, n), "rb") as f:
        b = f.read()
    try:
        pipeline.jpeg_info(b)
        blobs.append(b)
    except pipeline.I2sError:
        pass
blobs = blobs * 8
params = Params(contrast=70, brightness=50, schedule=True)
det = Detector(0, 16, 1300, 1300)
det.detect_jpeg(blobs[:16], params, full=False)
t = time.perf_counter()
b1 = det.detect_jpeg(blobs, params, full=False)
t_dev = time.perf_counter() - t
t = time.perf_counter()
imgs = [np.array(Image.open(io.BytesIO(b)).convert("RGB")) for b in blobs]
t_pil = time.perf_counter() - t
t = time.perf_counter()
b2 = det.detect_batch(imgs, params, full=False)
t_det = time.perf_counter() - t
assert all(bytes(x) == bytes(y) for x, y in zip(b1, b2))
mpx = sum(i.shape[0] * i.shape[1] for i in imgs) / 1e6
print("%d JPEGs, %.1f Mpx, %.1f MB of files" % (len(blobs), mpx, sum(len(b) for b in blobs) / 1e6))
print("Pillow decode (1 thread) %.1f ms + detect_batch %.1f ms = %.0f img/s" % (t_pil * 1e3, t_det * 1e3, len(blobs) / (t_pil + t_det)))
print("detect_jpeg (1 thread)   %.1f ms                      = %.0f img/s" % (t_dev * 1e3, len(blobs) / t_dev))
det.close()
from img2sgf_amd.pipeline import StreamedDetector      # noqa: E402
for n in (2, 4):
    sd = StreamedDetector(0, n, 16, 1300, 1300)
    sd.detect_jpeg(blobs[:16 * n], params)
    t = time.perf_counter()
    b3 = sd.detect_jpeg(blobs, params)
    dt = time.perf_counter() - t
    assert all(bytes(x) == bytes(y) for x, y in zip(b1, b3))
    sd.close()
    print("detect_jpeg, %d streams    %.1f ms                      = %.0f img/s" % (n, dt * 1e3, len(blobs) / dt))
