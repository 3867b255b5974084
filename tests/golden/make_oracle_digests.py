#!/usr/bin/env python3
"""Writes tests/golden/oracle_stage_digests.json: SHA-256 digests of what oracle/ (the CPU restatement of the reference's
OpenCV calls) answers at each of the ten call sites of img2sgf.py:153-198, 236-244 on the 18 reference fixtures (default
contrast / brightness) and three synthetic diagrams, plus -- for the three stages that SURVEY Appendix A.7's version switches
touch directly -- the digests under the alternative switch value.  Data only: inputs are identified by the digest of their pixels.

    python tests/golden/make_oracle_digests.py            (about two minutes of CPU)

Check on ANY machine with OpenCV (no GPU, no build of this repository):   python -m oracle.cv2_harness --digests
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import numpy as np  # noqa: E402

from oracle import cv_oracle as cvo, glue, pipeline as opipe, stage_digests as sd  # noqa: E402


def entry(img):
    r = opipe.process_image(img, compat=cvo.DIGEST_COMPAT)
    e = {"input": sd.sha(img, np.uint8), "shape": list(img.shape), "threshold": int(r["threshold"]), "stages": sd.stage_digests(r),
         "sgf": r["sgf"]}
    alt = {}
    if img.ndim == 3:
        alt["cvtColor:153@grey_shift=14"] = sd.sha(cvo.bgr2gray(img, 14), np.uint8)
    for k in (3, 5, 7):
        alt["gauss%d:174-175@gauss_kernel_mode=1" % k] = sd.sha(cvo.gaussian_blur(r["grey"], k, k, 1), np.uint8)
    for name, horizontal in (("HoughLines_H:236", True), ("HoughLines_V:240-247", False)):
        l = glue.find_lines(r["circles_removed"], r["threshold"], horizontal, 1)
        alt[name + "@houghlines_numangle=1"] = sd.sha(np.asarray(l, np.float32).reshape(-1), np.float32)
    e["alt"] = alt
    return e


def build(names=None):
    doc = {"what": "oracle/ outputs per OpenCV call site (img2sgf.py line numbers in the keys); sha256[:32] of str(shape) + bytes",
           "switches": dict(cvo.DIGEST_COMPAT), "inputs": {}}
    for name, img in sd.inputs(os.path.join(HERE, "test_images")):
        if names is None or name in names:
            doc["inputs"][name] = entry(img)
    return doc


if __name__ == "__main__":
    doc = build()
    with open(sd.DIGEST_FILE, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("wrote", sd.DIGEST_FILE, len(doc["inputs"]), "inputs")
