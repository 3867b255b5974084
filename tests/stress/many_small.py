import sys
sys.path.insert(0, ".")
import numpy as np
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params
rng = np.random.default_rng(5)
n = 4096
imgs = []
for k in range(n):
    h, w = int(rng.integers(1, 64)), int(rng.integers(1, 64))
    if k % 3 == 0:
        imgs.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
    elif k % 3 == 1:
        imgs.append(np.where(rng.random((h, w)) < 0.5, 0, 255).astype(np.uint8))
    else:
        imgs.append(rng.integers(0, 256, (h, w), dtype=np.uint8))
big = Detector(0, 4096, 64, 64)
small = Detector(0, 16, 64, 64)
a = big.detect_batch(imgs, full=False)
b = small.detect_batch(imgs, full=False)
bad = [k for k in range(n) if bytes(a[k]) != bytes(b[k])]
print("4096 tiny images in one pass vs passes of 16: differing records", len(bad), bad[:10])
big.close(); small.close()
