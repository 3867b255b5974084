import sys
sys.path.insert(0, ".")
import numpy as np, torch
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params
n = int(sys.argv[1])
dev, occs = synth.synth_batch_torch(range(n), torch.device("cuda", 0))
det = Detector(0, n, 1024, 1024)
boards = det.detect_device(dev, Params())
want, exc = synth.expected_boards(range(n), occs)
bad = [k for k in range(n) if not (np.ctypeslib.as_array(boards[k].board) == want[k]).all()]
print("pass of", n, "bad boards:", len(bad), bad[:8], "first bad >= 256:", all(k >= 256 for k in bad))
det.close()
