#!/usr/bin/env python3
"""CPU baseline of bench.py: the oracle (C restatement of the reference's OpenCV path + the reference's glue) on the
host cores, one single-threaded worker process per core, images split evenly (SURVEY 8d "P worker processes").

Runs in its own interpreter (bench.py starts it with subprocess) so that the workers can be forked without a HIP
runtime in the parent.  Prints one JSON object: {"value", "unit", "cores", "kind", "sample", "single_core_value"}.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(k, seeds, barrier, q):
    from img2sgf_amd import synth
    from oracle import pipeline as opipe
    imgs = [synth.synth_diagram(int(s))[0] for s in seeds]
    opipe.process_image(imgs[0], keep_planes=False)          # warm-up (page in the library, first-touch the buffers)
    barrier.wait()
    t0 = time.time()
    for im in imgs:
        opipe.process_image(im, keep_planes=False)
    q.put((k, t0, time.time(), len(imgs)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=0, help="worker processes (0 = one per host core, at most 64)")
    ap.add_argument("--per-worker", type=int, default=8, help="diagrams per worker")
    args = ap.parse_args()
    from oracle import cv_oracle
    cv_oracle.build()                                        # make sure liboracle exists before the workers race for it
    P = args.workers or min(os.cpu_count() or 1, 64)
    n = args.per_worker
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(P), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(k, range(k * n, (k + 1) * n), barrier, q)) for k in range(P)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join()
    t0, t1 = min(r[1] for r in res), max(r[2] for r in res)
    total = sum(r[3] for r in res)
    per_core = sum(r[3] / (r[2] - r[1]) for r in res) / P
    print(json.dumps(dict(
        value=total / (t1 - t0), unit="images/s", cores=P, kind="port",
        single_core_value=per_core,
        sample="%d synthetic 1024x1024 diagrams (seeds 0..%d), %d single-threaded worker processes x %d diagrams each; "
               "oracle/ C restatement of the reference's OpenCV path + the reference's glue (cv2 is not installed)"
               % (total, total - 1, P, n))))


if __name__ == "__main__":
    main()
