"""PCIe-inclusive rate of the benchmark workload: host numpy images (pageable memory) through i2s_detect_batch
(inputs_on_device = 0), one stream and several (StreamedDetector.detect_batch), for DESIGN.md."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from img2sgf_amd import synth                                            # noqa: E402
from img2sgf_amd.pipeline import Detector, Params, StreamedDetector      # noqa: E402

N = 1024
imgs, occs = synth.synth_batch(range(N))
lst = list(imgs)


def check(b):
    return all((np.ctypeslib.as_array(b[k].board) == occs[k]).all() for k in range(N))


det = Detector(0, 128, 1024, 1024)
det.detect_batch(lst[:128], Params(), full=False)
t0 = time.perf_counter()
b = det.detect_batch(lst, Params(), full=False)
dt = time.perf_counter() - t0
det.close()
print("host-input path, 1 stream : %.0f images/s (boards ok=%s)" % (N / dt, check(b)))
for n in (2, 3, 4):
    sd = StreamedDetector(0, n, 128, 1024, 1024)
    sd.detect_batch(lst[:128 * n], Params())
    t0 = time.perf_counter()
    b = sd.detect_batch(lst, Params())
    dt = time.perf_counter() - t0
    sd.close()
    print("host-input path, %d streams: %.0f images/s (boards ok=%s)" % (n, N / dt, check(b)))
