"""Per-stage SHA-256 digests of the reference's ten OpenCV calls  --  TEST INFRASTRUCTURE ONLY.

The oracle's answers on fixed inputs (the 18 reference fixtures after the reference's default contrast / brightness step, three
synthetic diagrams) are committed as digests (tests/golden/oracle_stage_digests.json, written by
tests/golden/make_oracle_digests.py).  Anybody with OpenCV -- no GPU, no build of this repository, only numpy + Pillow + cv2 --
can then confirm or refute the restatement:   python -m oracle.cv2_harness --digests
prints cv2.__version__, which of the SURVEY Appendix A.7 switch values the installed cv2 matches on the directly affected
stages, and every (input, stage) whose digest differs.  This module imports neither cv_oracle (the C library) nor cv2.
"""
import hashlib
import json
import os

import numpy as np

STAGE_NAMES = ["grey", "edges", "median1", "gauss1", "median3", "gauss3", "median5", "gauss5", "median7", "gauss7"]
DIGEST_FILE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "oracle_stage_digests.json")


def sha(a, dtype):
    a = np.ascontiguousarray(np.asarray(a, dtype))
    h = hashlib.sha256()
    h.update(str(a.shape).encode())
    h.update(a.tobytes())
    return h.hexdigest()[:32]


def stage_digests(r):
    """r: the dict run_cv2_calls / oracle.pipeline.process_image(keep_planes=True) return.  One digest per call site."""
    d = {"cvtColor:153": sha(r["grey"], np.uint8), "Canny:162": sha(r["edges"], np.uint8)}
    for k in range(2, 10):
        d["%s:174-175" % STAGE_NAMES[k]] = sha(r["blurs"][k], np.uint8)
    for k in range(10):
        d["HoughCircles(%s):180" % STAGE_NAMES[k]] = sha(np.asarray(r["circles_per_variant"][k], np.float32).reshape(-1, 3), np.float32)
    d["circles:186"] = sha(np.asarray(r["circles_all"], np.float32).reshape(-1, 3), np.float32)
    d["erase:191-198"] = sha(r["circles_removed"], np.uint8)
    d["HoughLines_H:236"] = sha(np.asarray(r["hlines"], np.float32).reshape(-1), np.float32)
    d["HoughLines_V:240-247"] = sha(np.asarray(r["vlines"], np.float32).reshape(-1), np.float32)
    return d


def inputs(fixture_dir):
    """(name, input_image_np) of the digest set: the reference's 18 test images as the reference holds them at img2sgf.py:150
    with default settings (Pillow decode, contrast 70, brightness 50), and synthetic 1024 x 1024 diagrams (seeds 0, 1; seed 0 noisy)."""
    from . import pipeline as opipe
    from img2sgf_amd import synth
    for n in sorted(os.listdir(fixture_dir)):
        if n.endswith(".jpg"):
            yield n, opipe.load_and_enhance(os.path.join(fixture_dir, n))
    for seed in (0, 1):
        yield "synth%d" % seed, synth.synth_diagram(seed)[0]
    yield "synth_noisy0", synth.synth_diagram(0, noisy=True)[0]


def load(path=None):
    with open(path or DIGEST_FILE) as f:
        return json.load(f)
