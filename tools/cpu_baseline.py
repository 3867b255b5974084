#!/usr/bin/env python3
"""CPU baseline of bench.py, on a bounded sample of the benchmark workload.

If `cv2` is importable (it is the reference's own arithmetic), the per-image path is the reference's ten cv2 calls +
glue exactly as img2sgf.py:153-198, 236-244 (HoughLines pairs executed twice, as the reference does at :269)
-- oracle/cv2_harness.py:cv2_process_image -- and two figures are reported:
  B1  one process, OpenCV's default thread pool (cv2.setNumThreads(0) semantics = library default);
  B2  P = os.cpu_count() worker processes with cv2.setNumThreads(1) each, images split evenly;  -> `value`, kind "cv2".
Otherwise the oracle (oracle/: C restatement of the same OpenCV path + the reference's glue) stands in, kind "port":
  B1' one single-threaded process alone;  B3  P single-threaded processes -> `value`.

P = the CPUs this process may actually use: min(len(os.sched_getaffinity(0)), cgroup CPU quota (cpu.max / cfs_quota)), NOT
os.cpu_count() (round 2 reported 256 "cores" on a box whose job ran 8.6x faster on them than on one).  The output states
`host_cores`, `affinity_cores`, `cgroup_quota_cores`, `cores` (= P) and a self-check: `scaling` = value / b1.value and
`scaling_suspect` = scaling < 0.5 * P.  The port's buffers: glibc is told (mallopt) to keep freed multi-megabyte blocks on the
heap instead of returning them to the kernel with munmap, so that after the warm-up image a worker measures arithmetic, not
page faults.

Runs in its own interpreter (bench.py starts it with subprocess) so that the workers can be forked without a HIP
runtime in the parent.  Prints one JSON object:
  {"value", "unit", "cores", "kind", "sample", "b1": {"value", "threads", ...}, "cpu_model", ...}.
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cgroup_quota():
    """CPUs the cgroup's bandwidth controller allows (None = unlimited / unknown)."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:               # cgroup v2: "<quota> <period>" or "max <period>"
            q, per = f.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:  # cgroup v1
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def effective_cores():
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = cgroup_quota()
    p = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return p, aff, quota


def keep_heap():
    """glibc: no mmap / munmap per multi-megabyte buffer (the port allocates its planes per call)."""
    try:
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)        # M_MMAP_THRESHOLD
        libc.mallopt(-1, (1 << 31) - 1)  # M_TRIM_THRESHOLD
        libc.mallopt(-2, 64 << 20)       # M_TOP_PAD
    except OSError:
        pass


def per_image_fn(use_cv2, threads):
    keep_heap()
    if use_cv2:
        import cv2
        from oracle import cv2_harness
        cv2.setNumThreads(threads)            # 0 = OpenCV's default pool (B1); 1 = single-threaded worker (B2)
        return cv2_harness.cv2_process_image
    from oracle import pipeline as opipe
    return lambda im: opipe.process_image(im, keep_planes=False)


def worker(k, seeds, barrier, q, use_cv2):
    from img2sgf_amd import synth
    fn = per_image_fn(use_cv2, 1)
    imgs = [synth.synth_diagram(int(s))[0] for s in seeds]
    fn(imgs[0])                                               # warm-up (page in the library, first-touch the buffers)
    barrier.wait()
    t0 = time.time()
    for im in imgs:
        fn(im)
    q.put((k, t0, time.time(), len(imgs)))


def single_process(use_cv2, n, repeats=3):
    """B1: one process (cv2: default thread pool), warm-up 1 image, median of `repeats` passes over n images."""
    from img2sgf_amd import synth
    fn = per_image_fn(use_cv2, 0)
    imgs = [synth.synth_diagram(s)[0] for s in range(n)]
    fn(imgs[0])
    rates = []
    for _ in range(repeats):
        t0 = time.time()
        for im in imgs:
            fn(im)
        rates.append(n / (time.time() - t0))
    rates.sort()
    threads = 1
    if use_cv2:
        import cv2
        threads = cv2.getNumThreads()
    return dict(value=rates[len(rates) // 2], unit="images/s", threads=threads, images=n, repeats=repeats)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workers", type=int, default=0, help="worker processes (0 = one per CPU this process may use: affinity and cgroup quota)")
    ap.add_argument("--per-worker", type=int, default=8, help="diagrams per worker")
    ap.add_argument("--b1-images", type=int, default=4, help="diagrams of the single-process leg")
    ap.add_argument("--force-port", action="store_true", help="use the oracle even if cv2 is importable")
    args = ap.parse_args()
    from oracle import cv_oracle, cv2_harness
    cv_oracle.build()                                        # make sure liboracle exists before the workers race for it
    use_cv2 = cv2_harness.have_cv2() and not args.force_port
    p_eff, aff, quota = effective_cores()
    P = args.workers or p_eff
    n = args.per_worker
    b1 = single_process(use_cv2, args.b1_images)
    ctx = mp.get_context("fork")
    barrier, q = ctx.Barrier(P), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(k, range(k * n, (k + 1) * n), barrier, q, use_cv2)) for k in range(P)]
    for p in procs:
        p.start()
    res = [q.get() for _ in procs]
    for p in procs:
        p.join()
    t0, t1 = min(r[1] for r in res), max(r[2] for r in res)
    total = sum(r[3] for r in res)
    per_core = sum(r[3] / (r[2] - r[1]) for r in res) / P
    if use_cv2:
        what = ("the reference's ten cv2 calls + glue per image (img2sgf.py:153-198, 236-244, HoughLines pairs twice), cv2 %s; "
                "B2 = %d processes x cv2.setNumThreads(1), B1 = 1 process with OpenCV's default pool (%d threads)"
                % (cv2_harness.cv2_version(), P, b1["threads"]))
    else:
        what = ("oracle/ C restatement of the reference's OpenCV path + the reference's glue (cv2 is not installed on this box); "
                "B3 = %d single-threaded processes, B1' = 1 process alone" % P)
    value = total / (t1 - t0)
    scaling = value / b1["value"] if b1["value"] > 0 else None
    print(json.dumps(dict(
        value=value, unit="images/s", cores=P, kind="cv2" if use_cv2 else "port",
        per_core_value=per_core, b1=b1, cpu_model=cpu_model(), host_cores=os.cpu_count(),
        affinity_cores=aff, cgroup_quota_cores=quota, cores_effective=p_eff,
        scaling=scaling, scaling_suspect=bool(scaling is not None and not use_cv2 and scaling < 0.5 * P),
        sample="%d synthetic 1024x1024 diagrams (seeds 0..%d), %d worker processes x %d diagrams each; %s"
               % (total, total - 1, P, n, what))))


if __name__ == "__main__":
    main()
