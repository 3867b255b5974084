import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from img2sgf_amd.pipeline import Detector, Params
from img2sgf_amd import preprocess
p = "tests/golden/test_images/ex1.jpg"
raw = np.array(preprocess.load_image(p))
want = preprocess.enhance(preprocess.load_image(p), 70, 50)
det = Detector(0, 1, raw.shape[1], raw.shape[0])
det.detect_batch([raw], Params(contrast=70, brightness=50), full=False)
got = det.fetch_source(0)
bad = np.argwhere(got != want)
print(raw.shape, len(bad), bad[:5].tolist())
for b in bad[:8]:
    print(tuple(b), "raw", raw[tuple(b)], "got", got[tuple(b)], "want", want[tuple(b)])
r, g, bb = [raw[..., i].astype(np.int64) for i in range(3)]
L = (r * 19595 + g * 38470 + bb * 7471 + 0x8000) >> 16
print("mean", L.sum() / L.size, int(L.sum() / L.size + 0.5))
# infer mean used by the GPU from a mid-range pixel
d = (got.astype(int) - want.astype(int)); print("diff hist", np.unique(d, return_counts=True))
