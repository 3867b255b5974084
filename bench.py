#!/usr/bin/env python3
"""Benchmark of the board-detection hot path: go-diagram images/sec (1024x1024 greyscale) on N MI355X.

A step = one pass of the whole hot path (grey -> blur bank -> Canny -> 10x HoughCircles -> erase -> HoughLines ->
grid -> classifier) over one batch of synthetic diagrams resident in HBM, per rank; ranks own disjoint seed ranges
(no data-path collective), then all-gather the 384-byte board records over RCCL.  Prints one JSON line.

Workload (BASELINE.json configs[2]): 4096 synthetic 1024x1024 19x19 diagrams per GPU, seeds rank*4096 .. +4095.
The timed region drives `--streams` HIP streams per GPU (independent contexts; the latency-bound tail kernels of one
slice overlap the throughput-bound kernels of another).  The roofline object is measured separately, right after the
timed region, on ONE stream (concurrent streams would stretch every per-kernel duration): HIP events recorded on the
context's stream around the blur+Canny stage, `--roofline-images` diagrams of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: measured copy ceiling
N_PIX = 1024 * 1024
BLUR_CANNY_BYTES = 14 * N_PIX  # Canny 2N + 3 Gaussians 6N + 3 medians 6N, unfused accounting


def cpu_baseline(per_worker):
    """The oracle (CPU restatement of the reference's OpenCV path + the reference's glue) on the host cores: one
    single-threaded worker process per core (tools/cpu_baseline.py, run in its own interpreter so that the workers fork
    without a HIP runtime in the parent), on a bounded sample of the same workload."""
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--per-worker", str(per_worker)],
                         stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, check=True, timeout=900).stdout
    return json.loads(out.decode().strip().splitlines()[-1])


def measured_traffic():
    """HBM bytes per image of the blur+Canny stage from the committed PMC passes (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        with open(p) as f:
            return json.load(f).get("blur_canny_hbm_bytes_per_image")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("I2S_BENCH_BATCH", 4096)), help="diagrams per rank per step")
    ap.add_argument("--pass-size", type=int, default=int(os.environ.get("I2S_BENCH_PASS", 128)), help="diagrams per device pass")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("I2S_BENCH_STREAMS", 3)), help="HIP streams (contexts) per GPU")
    ap.add_argument("--roofline-images", type=int, default=256)
    ap.add_argument("--cpu-per-worker", type=int, default=8, help="diagrams per CPU worker process in the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from img2sgf_amd import synth, dist as i2s_dist
    from img2sgf_amd.pipeline import Detector, Params, StreamedDetector

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    use_dist = world > 1 or "RANK" in os.environ        # launched by torch.distributed.run: always go through RCCL
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    torch.cuda.set_device(local)
    B = args.batch
    lo, hi = i2s_dist.shard_range(B * world, rank, world)
    dev, occs = synth.synth_batch_torch(range(lo, hi), torch.device("cuda", local))   # rendered on the GPU, resident in HBM
    torch.cuda.synchronize()
    pass_size = min(args.pass_size, B)
    if args.streams > 1:
        det = StreamedDetector(local, args.streams, pass_size, 1024, 1024)
    else:
        det = Detector(local, pass_size, 1024, 1024)
    params = Params()

    def step():
        boards = det.detect_device(dev, params)
        return i2s_dist.allgather_boards(boards, world, local)

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        allb = step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    # sanity: the boards this rank produced equal the generator's occupancy
    mine = allb[lo:hi]
    ok = bool((mine[:, :361].reshape(-1, 19, 19) == occs).all())
    det.close()

    out = None
    if rank == 0:
        # roofline of the blur+Canny stage: one stream, HIP events around the stage on that stream
        nr = min(args.roofline_images, B)
        d1 = Detector(local, min(pass_size, nr), 1024, 1024)
        d1.detect_device(dev[:nr], params)
        d1.detect_device(dev[:nr], params)
        timing = d1.last_timing()
        d1.close()
        stage_s = timing["blur_canny_ms"] * 1e-3
        ach = BLUR_CANNY_BYTES * nr / stage_s / 1e9
        traffic = measured_traffic()
        images = B * world * args.steps
        out = {
            "metric": "go-diagram images/sec (1024x1024 greyscale)", "value": images / dt, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic 1024x1024 19x19 diagrams per GPU (BASELINE configs[2]), "
                                   "device-resident, full hot path incl. board all-gather" % B,
                       "pass_size": pass_size, "streams": args.streams, "boards_match_generator": ok},
            "roofline": {"bound": "hbm",
                         "kernel": "blur+Canny stage: k_grey, k_median3, k_median57, k_gauss357, k_sobel_nms_planes(main), "
                                   "k_hysteresis(map 0); the main-Canny pass also emits HoughCircles' Canny map of the grey plane",
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "frac_of_measured_copy_ceiling": ach / HBM_COPY_GBS,
                         "traffic": traffic, "algorithmic_bytes_per_image": BLUR_CANNY_BYTES,
                         "stage_us_per_image": stage_s / nr * 1e6,
                         "measured_on": "1 stream, %d diagrams, HIP events on the context's stream" % nr},
            "single_stream_stage_ms": timing,
        }
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_per_worker)
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
