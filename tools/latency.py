"""Latency of ONE call -- the reference's interactive use (its Tk loop calls process_image once per slider move, img2sgf.py:1077-1191;
BASELINE configs[1]): host numpy image in, board record / full Detection out, for a 1024 x 1024 synthetic diagram and for the reference's
ex1.jpg after its default contrast step; batch sizes 1, 4, 16 for comparison.  Also the sum of the call's kernel durations
(i2s_last_kernel_timing) -- what is left is launches, memsets, copies and the final synchronisation.  For profiles/HISTORY.md 6c."""
import os
import statistics
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from img2sgf_amd import preprocess, synth                    # noqa: E402
from img2sgf_amd.pipeline import Detector, Params            # noqa: E402

REPS = 60


def measure(det, imgs, full):
    p = Params()
    for _ in range(5):
        det.detect_batch(imgs, p, full=full)
    ts = []
    for _ in range(REPS):
        t0 = time.perf_counter()
        det.detect_batch(imgs, p, full=full)
        ts.append(time.perf_counter() - t0)
    det.set_profiling(True)
    det.detect_batch(imgs, p, full=full)
    k = sum(det.last_kernel_timing().values())
    det.set_profiling(False)
    return statistics.median(ts) * 1e3, min(ts) * 1e3, k


def main():
    diag, _ = synth.synth_batch(range(16))
    ex1 = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "test_images", "ex1.jpg")
    scan = np.ascontiguousarray(preprocess.enhance(preprocess.load_image(ex1), 70, 50))
    print("one call of i2s_detect_batch, host arrays in -> records out; ms: median / best of %d, kernels = sum of kernel durations" % REPS)
    for name, imgs in (("1024x1024 diagram", list(diag)), ("ex1.jpg (%dx%d RGB)" % (scan.shape[1], scan.shape[0]), [scan] * 16)):
        for nb in (1, 4, 16):
            det = Detector(0, nb, max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
            for full in (False, True):
                med, best, k = measure(det, imgs[:nb], full)
                print("  %-24s batch %2d  %-12s  %7.3f / %7.3f ms   kernels %7.3f ms   (%.3f ms per image)"
                      % (name, nb, "full record" if full else "board only", med, best, k, med / nb))
            det.close()


if __name__ == "__main__":
    main()
