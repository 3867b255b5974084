import sys, time
sys.path.insert(0, "."); sys.path.insert(0, "tests")   # run from the repo root
import numpy as np
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params, StreamedDetector
from oracle import pipeline as opipe
import os
rng = np.random.default_rng(2)
fix = [opipe.load_and_enhance(os.path.join("tests/golden/test_images", n)) for n in sorted(os.listdir("tests/golden/test_images"))]
noisy = [synth.synth_diagram(s, noisy=True)[0] for s in range(6)]
snake = np.full((1000, 1200), 100, np.uint8)
for y in range(20, 980, 24): snake[y:y + 12, 10:1190] = 130
for k, y in enumerate(range(20, 956, 24)):
    x = 1170 if k % 2 == 0 else 10
    snake[y:y + 36, x:x + 20] = 130
snake[20:32, 10:14] = 255
imgs = (fix + noisy + [snake, snake.T.copy()]) * 4
order = rng.permutation(len(imgs)); imgs = [imgs[i] for i in order]
one = Detector(0, 8, 1300, 1300)
want = one.detect_batch(imgs, full=False)
print("single context:", one.hysteresis_stats())
for ns in (2, 4, 6):
    sd = StreamedDetector(0, ns, 8, 1300, 1300)
    worst = 0
    for rep in range(6):
        t0 = time.perf_counter()
        got = sd.detect_batch(imgs)
        dt = time.perf_counter() - t0
        worst = max(worst, dt)
        bad = [k for k in range(len(imgs)) if bytes(got[k]) != bytes(want[k])]
        if bad: print("MISMATCH streams", ns, "rep", rep, bad[:10])
    print(ns, "streams: worst call %.1f ms" % (worst * 1e3), [d.hysteresis_stats() for d in sd.dets])
    for d in sd.dets: d.close()
