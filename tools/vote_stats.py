#!/usr/bin/env python3
"""Vote count of the HoughCircles stage on the benchmark workload ("K5's vote stage is bound by LDS atomic
throughput -- report votes/s").  Sums the debug accumulators of all 8 variants for a few synthetic diagrams on the GPU and
prints the average votes per image; divide by the per-image duration of k_vote_centres (profiles/) for votes/s.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    from img2sgf_amd import synth
    from img2sgf_amd.pipeline import Detector
    imgs, _ = synth.synth_batch(range(n))
    det = Detector(0, n, 1024, 1024)
    det.set_debug(True)
    dets = det.detect_batch(list(imgs), None, full=True)
    votes = [[int(det.fetch_circle_acc(i, v).sum()) for v in range(8)] for i in range(n)]
    det.close()
    per_img = [sum(v) for v in votes]
    print(json.dumps(dict(images=n, votes_per_image=sum(per_img) / n, votes_per_variant_image0=votes[0],
                          circles_image0=int(len(dets[0].circles_all)) if hasattr(dets[0], "circles_all") else None)))


if __name__ == "__main__":
    main()
