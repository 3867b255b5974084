#!/usr/bin/env python3
"""Writes tests/golden/fixture_outcomes.json: what the ORACLE (oracle/, the CPU restatement of the reference's OpenCV calls plus the
reference-pinned glue) makes of the reference's 18 fixtures at the reference's default settings (img2sgf.py:616-640: contrast 70,
brightness 50, no rotation, whole image, threshold = choose_threshold) under every named set of the OpenCV-version switches
(tests/switches.py): circles found, line clusters, board size, stone counts, side to move and the SGF text.

Data, not source: inputs are the committed fixtures, outputs are what the restatement answers.  It is the evidence table for
DESIGN.md section 2a ("which OpenCV do the defaults restate") and the expected values of tests/test_fixture_outcomes.py, where
the HIP path has to reproduce every entry through the C ABI.

    python tests/golden/make_fixture_outcomes.py            (about a minute of CPU)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import switches  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402

OUT = os.path.join(HERE, "fixture_outcomes.json")
IMAGES = ["ex%d.jpg" % i for i in range(1, 18)] + ["no_circles.jpg"]


def outcome(img, sw):
    r = opipe.process_image(img, compat=switches.compat(sw), keep_planes=False)
    o = {"threshold": int(r["threshold"]), "n_circles": int(len(r["circles_all"])),
         "n_hlines": int(len(r["hlines"])), "n_vlines": int(len(r["vlines"])),
         "n_hclusters": int(len(r["hcentres"])), "n_vclusters": int(len(r["vcentres"])),
         "found_grid": bool(r["found_grid"]), "valid_grid": bool(r["valid_grid"]), "board_ready": bool(r["board_ready"]),
         "hsize": int(r["hsize"]), "vsize": int(r["vsize"]), "sgf": r["sgf"]}
    if r["board_ready"]:
        o.update(n_black=int(r["num_black_stones"]), n_white=int(r["num_white_stones"]), side_to_move=int(r["side_to_move"]))
    return o


def build(names=IMAGES, sets=switches.NAMES):
    doc = {"what": "oracle outcomes of the reference's fixtures at its default settings, per OpenCV-version switch set",
           "switch_sets": {s: switches.compat(s) for s in sets}, "fixtures": {}}
    for n in names:
        img = opipe.load_and_enhance(os.path.join(HERE, "test_images", n))
        doc["fixtures"][n] = {s: outcome(img, s) for s in sets}
    return doc


if __name__ == "__main__":
    doc = build()
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
    print("%-15s" % "fixture" + "".join("%-22s" % s for s in switches.NAMES))
    for n, per in doc["fixtures"].items():
        cells = []
        for s in switches.NAMES:
            o = per[s]
            cells.append("%dx%d %dB/%dW" % (o["hsize"], o["vsize"], o["n_black"], o["n_white"]) if o["board_ready"]
                         else "no board (%dx%d cl)" % (o["n_vclusters"], o["n_hclusters"]))
        print("%-15s" % n + "".join("%-22s" % c for c in cells))
