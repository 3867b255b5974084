#!/usr/bin/env python3
"""Seeds of the synthetic workload whose recovered board differs from the generator's occupancy (GPU path; the oracle has to
agree on every one of them: tests/golden/make_synth_exceptions.py).  usage: tools/synth_mismatches.py [first] [count]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
    import numpy as np
    import torch
    from img2sgf_amd import dist as i2s_dist, synth
    from img2sgf_amd.pipeline import Params, StreamedDetector
    sd = StreamedDetector(0, 3, 256, 1024, 1024)
    bad = []
    for lo in range(first, first + count, 4096):
        hi = min(lo + 4096, first + count)
        dev, occs = synth.synth_batch_torch(range(lo, hi), torch.device("cuda", 0))
        t = i2s_dist.boards_to_numpy(sd.detect_device(dev, Params()))
        miss = np.nonzero((t[:, :361].reshape(-1, 19, 19) != occs).any(axis=(1, 2)))[0]
        bad += [int(lo + k) for k in miss]
        del dev
    sd.close()
    print(json.dumps({"first": first, "count": count, "mismatching_seeds": bad}))


if __name__ == "__main__":
    main()
