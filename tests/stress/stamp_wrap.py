import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")   # run from the repo root
import numpy as np
from img2sgf_amd.pipeline import Detector, Params
rng = np.random.default_rng(8)
img = np.clip(rng.normal(120, 40, (70, 90)), 0, 255).astype(np.uint8)      # plenty of weak pixels
det = Detector(0, 1, 90, 70)
first = bytes(det.detect_batch([img], full=False)[0])
e0 = det.fetch_plane(0, "edges").copy()
ptr = ([img.ctypes.data], [90], [70], [90], [1])
p = Params()
t0 = time.perf_counter()
n = 290000
bad = 0
for i in range(n):
    b, _ = det.detect_ptrs(*ptr, p, False)
    if bytes(b[0]) != first:
        bad += 1
        if bad < 5: print("record differs at call", i)
    if i % 20000 == 0 or i > 262000 and i % 1000 == 0:
        if not np.array_equal(det.fetch_plane(0, "edges"), e0):
            bad += 1; print("edges differ at call", i)
print("%d calls in %.1f s, mismatches %d" % (n, time.perf_counter() - t0, bad), det.hysteresis_stats())
