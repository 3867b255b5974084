"""PCIe-inclusive rate: host numpy images through i2s_detect_batch (inputs_on_device = 0), for DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params
imgs, occs = synth.synth_batch(range(256))
det = Detector(0, 64, 1024, 1024)
lst = list(imgs)
det.detect_batch(lst, Params(), full=False)
t0 = time.perf_counter()
for _ in range(3):
    b = det.detect_batch(lst, Params(), full=False)
dt = time.perf_counter() - t0
ok = all((np.ctypeslib.as_array(b[k].board) == occs[k]).all() for k in range(256))
print("host-input path: %.0f images/s (1 stream, pageable host memory, boards ok=%s)" % (3 * 256 / dt, ok))
