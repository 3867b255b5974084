import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from img2sgf_amd.pipeline import Detector
from oracle import pipeline as opipe, cv_oracle as cvo
import parity
rng = np.random.default_rng(5)
imgs = [rng.integers(0, 256, (h, w), dtype=np.uint8) for (h, w) in [(1, 1), (2, 5), (7, 3), (33, 65), (130, 129)]]
for idx in (3, 4):
    img = imgs[idx]
    for trial in range(3):
        det = Detector(0, 1, 330, 300)
        det.set_debug(True)
        d = det.detect_batch([img])[0]
        ref = opipe.process_image(img)
        for name, want in zip(parity.VARIANT_PLANES, parity.oracle_variants(ref)):
            got = det.fetch_plane(0, name)
            if (got != want).any():
                bad = np.argwhere(got != want); print(idx, trial, name, "differs", len(bad), bad[:5].tolist())
        got = det.fetch_plane(0, "removed")
        if (got != ref["circles_removed"]).any(): print(idx, trial, "removed differs")
        e, m = cvo.canny(img, 50, 200, return_map=True)
        gm = det.fetch_plane(0, 9)
        print(idx, trial, "map0 edges equal:", ((gm == 2) == (e == 255)).all(), "hlines equal:", np.array_equal(d.hlines, ref["hlines"]), "vlines equal:", np.array_equal(d.vlines, ref["vlines"]))
        acc, nang = det.fetch_line_acc(0)
        h, w = img.shape
        import math
        dd = math.pi/180
        _, dbg = cvo.hough_lines(ref["circles_removed"], 1, dd, ref["threshold"], math.pi/2-dd, math.pi/2+dd, debug=True)
        oa = dbg["acc"][1:1+nang[0], 1:-1]
        ga = acc[0:nang[0]]
        if not np.array_equal(oa, ga):
            bad = np.argwhere(oa != ga); print("  H acc differs at", len(bad), bad[:6].tolist(), [ (int(oa[tuple(b)]), int(ga[tuple(b)])) for b in bad[:6]])
        det.close()
