#!/usr/bin/env python3
"""Benchmark of the board-detection hot path: go-diagram images/sec (1024x1024 greyscale) on N MI355X.

A step = one pass of the whole hot path (grey -> blur bank -> Canny -> 10x HoughCircles -> erase -> HoughLines ->
grid -> classifier) over one batch of synthetic diagrams resident in HBM, per rank; ranks own disjoint seed ranges
(no data-path collective); the only exchange is one all-gather of the 384-byte board records.  Prints one JSON line.

Launch: `python bench.py --gpus N` starts the N ranks itself (re-executes under `python -m torch.distributed.run
--nproc-per-node N`, rendezvous on 127.0.0.1); started by torch.distributed.run directly (RANK / WORLD_SIZE in the
environment) it is one rank of that job.  With ranks, the records are all-gathered DEVICE TO DEVICE over RCCL through
the C ABI (`i2s_comm_create` / `i2s_set_board_sink` / `i2s_allgather_boards`): detect calls leave their records in the
rank's shard of the gather buffer on the GPU, `ncclAllGather` runs in place on the context's stream, one D2H copy
delivers the table.  `n_gpus` in the output is the number of ranks RCCL saw.

Workload (BASELINE.json configs[2]; configs[3] at --gpus 8): 4096 synthetic 1024x1024 19x19 diagrams per GPU, seeds
rank*4096 .. +4095.  The timed region drives `--streams` HIP streams per GPU (independent contexts; the latency-bound
tail kernels of one slice overlap the throughput-bound kernels of another).  The roofline objects are measured
separately, right after the timed region, on ONE stream (concurrent streams would stretch every per-kernel duration):
a pair of HIP events per kernel launch on the context's stream, stamped with the kernel's own start and stop (hipExtLaunchKernelGGL: the
duration a rocprofv3 kernel trace reports), `--roofline-images` diagrams of the same workload.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_GBS = 6290.0          # same guide: measured copy ceiling
N_PIX = 1024 * 1024
BLUR_CANNY_BYTES = 14 * N_PIX  # Canny 2N + 3 Gaussians 6N + 3 medians 6N, unfused accounting
# conflict-free LDS-atomic ceiling of one MI355X (profiles/r01_g_lds_atomic_microbench.txt): 13.5 lanes per CU-cycle
LDS_ATOMIC_PEAK = 13.5 * 256 * 2.4e9
BLUR_CANNY_SEGS = ("k_grey", "k_median3", "k_median57_bin", "k_median57", "k_gauss357", "k_blur", "k_sobel_nms(main Canny)",
                   "k_hysteresis(main Canny)")
CANNY7_SEGS = ("k_sobel_nms_rows(HoughCircles x7)", "k_hysteresis(HoughCircles)")
CANNY7_BYTES = 7 * 2 * N_PIX   # HoughCircles' internal Canny of the 7 planes that are not the grey plane: read N + write N each
NOISE_SIGMA = 6.0              # SURVEY 8(d) config 2, variant "noisy": N(0, 6^2) added to the same diagrams, clipped


def fixtures_roofline(device, copies=16):
    """The blur+Canny stage on the reference's OWN inputs: the 18 fixtures (tests/golden/test_images: data, committed) after the
    reference's default contrast / brightness step (Pillow on the host, img2sgf.py:136-150), `copies` times, one device pass, one
    stream, HIP start / stop events of every kernel launch.  Real scans are 88-97 % pure black / white PIXELS, but anti-aliased strokes touch
    nearly every 256 x 64 band, so the two-valued speculation of k_blur rarely holds: this, not the synthetic diagrams' figure,
    is what the stage does on a user's files.  Bytes: 14 per processed pixel (SURVEY 8d's unfused accounting, C = 1)."""
    from img2sgf_amd import preprocess
    from img2sgf_amd.pipeline import Detector, Params
    d = os.path.join(ROOT, "tests", "golden", "test_images")
    names = sorted(n for n in os.listdir(d) if n.endswith(".jpg")) if os.path.isdir(d) else []
    if not names:
        return None
    imgs = [np.ascontiguousarray(preprocess.enhance(preprocess.load_image(os.path.join(d, n)), 70, 50)) for n in names] * copies
    det = Detector(device, len(imgs), max(i.shape[1] for i in imgs), max(i.shape[0] for i in imgs))
    det.set_profiling(True)
    det.detect_batch(imgs, Params(), full=False)
    det.detect_batch(imgs, Params(), full=False)
    seg = det.last_kernel_timing()
    flagged, total = det.blur_band_stats()
    hy = det.hysteresis_stats()
    det.close()
    pixels = sum(i.shape[0] * i.shape[1] for i in imgs)
    stage_s = sum(seg.get(k, 0.0) for k in BLUR_CANNY_SEGS) * 1e-3
    ach = 14.0 * pixels / stage_s / 1e9
    return {"bound": "hbm", "kernel": "blur+Canny stage on the reference's 18 fixtures x %d (default contrast / brightness), ragged sizes, one pass" % copies,
            "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes": 14 * pixels, "pixels": pixels, "stage_us": stage_s * 1e6,
            "bands_not_two_valued": flagged, "bands": total, "bands_not_two_valued_frac": flagged / max(total, 1),
            "kernel_us": {k: seg[k] * 1e3 for k in BLUR_CANNY_SEGS if k in seg},
            # how often a real scan's hysteresis makes the host run a device pass again (i2s_hysteresis_stats)
            "device_passes": hy["passes"], "device_passes_redone": hy["redone"], "hysteresis_passes_max": hy["used_max"],
            "whole_path_images_per_s_single_stream": len(imgs) / (sum(seg.values()) * 1e-3)}


def cpu_baseline(per_worker):
    """tools/cpu_baseline.py in its own interpreter (its workers fork; no HIP runtime in their parent): the reference's
    ten cv2 calls + glue when cv2 is importable (kind "cv2": B1 = 1 process with OpenCV's own thread pool, B2 = one
    single-threaded process per core), otherwise the oracle (kind "port") on every host core; bounded sample."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--per-worker", str(per_worker)],
                         stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, check=True, timeout=1200).stdout
    return json.loads(out.decode().strip().splitlines()[-1])


# the sources of the blur+Canny stage's kernels (and what they include)
STAGE_SOURCES = ("i2s_types.h", "k_canny.h", "k_canny_rows.h", "k_filters.h", "tile_io.h", "isa/gfx950_ops.h")


def kernels_sha():
    """Hash of the kernel sources: PMC traffic figures are only valid for the kernels they were collected on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "img2sgf_amd", "csrc")
    for f in STAGE_SOURCES:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def measured_traffic():
    """(HBM bytes per image of the blur+Canny stage, provenance) from the committed PMC passes (profiles/traffic.json,
    written by tools/pmc_traffic.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this command).
    None -- not a stale number -- when the kernels changed since the counters were collected."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(p):
        return None, None, {"file": None}
    with open(p) as f:
        t = json.load(f)
    src = {"file": "profiles/traffic.json", "kernels_sha_of_counters": t.get("kernels_sha"), "kernels_sha_now": kernels_sha()}
    if t.get("kernels_sha") != src["kernels_sha_now"]:
        src["stale"] = True
        return None, None, src
    return t.get("blur_canny_hbm_bytes_per_image"), t.get("canny7_hbm_bytes_per_image"), src


def self_launch(args):
    """`python bench.py --gpus N` with no rank environment: start the N ranks (one per GPU) and pass their output on."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("I2S_BENCH_BATCH", 4096)), help="diagrams per rank per step")
    ap.add_argument("--pass-size", type=int, default=int(os.environ.get("I2S_BENCH_PASS", 256)), help="diagrams per device pass")
    ap.add_argument("--streams", type=int, default=int(os.environ.get("I2S_BENCH_STREAMS", 3)), help="HIP streams (contexts) per GPU")
    ap.add_argument("--roofline-images", type=int, default=256)
    ap.add_argument("--cpu-per-worker", type=int, default=8, help="diagrams per CPU worker process in the cpu_baseline leg")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-fixtures", action="store_true", help="skip the roofline_fixtures leg (the stage on the reference's 18 scans)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    from img2sgf_amd import synth, dist as i2s_dist
    from img2sgf_amd.pipeline import Detector, Params, StreamedDetector

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    use_dist = "RANK" in os.environ                     # a rank of a torch.distributed.run job: always go through RCCL
    if use_dist and world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting the ranks that exist" % (args.gpus, world),
              file=sys.stderr)
    gather = None
    torch.cuda.set_device(local)
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        world = dist.get_world_size()
        # the library's own RCCL communicator for the board records: rank 0's unique id travels over torch's
        uid = torch.zeros(128, dtype=torch.uint8, device="cuda")
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(i2s_dist.BoardGather.unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, 0)
        gather = i2s_dist.BoardGather(local, world, rank, args.batch * world, bytes(uid.cpu().numpy().tobytes()))
    B = args.batch
    total = B * world
    lo, hi = i2s_dist.shard_range(total, rank, world)
    dev, occs = synth.synth_batch_torch(range(lo, hi), torch.device("cuda", local))   # rendered on the GPU, resident in HBM
    torch.cuda.synchronize()
    pass_size = min(args.pass_size, B)
    if args.streams > 1:
        det = StreamedDetector(local, args.streams, pass_size, 1024, 1024)
        det0 = det.dets[0]
    else:
        det = det0 = Detector(local, pass_size, 1024, 1024)
    params = Params()

    def step():
        if gather is None:
            return i2s_dist.boards_to_numpy(det.detect_device(dev, params)).copy()
        det.detect_device(dev, params, sink=gather.sink(0))       # records stay on the device, in this rank's shard
        return gather.allgather(det0)                             # ncclAllGather in place + one D2H of the whole table

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
    # sanity: every rank holds the whole table, and the boards this rank produced equal the generator's occupancy
    assert allb.shape == (total, 384)
    mine = allb[lo:hi]
    # ... or, for the handful of seeds whose diagram the reference's algorithm itself reads differently (synth.algorithm_exceptions:
    # 15634 is the only one below 32768), the algorithm's answer
    want, exceptions = synth.expected_boards(range(lo, hi), occs, params.switch_set())
    ok = bool((mine[:, :361].reshape(-1, 19, 19) == want).all())
    if use_dist:
        t = torch.tensor([1 if ok else 0], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item())
    det.close()

    if rank == 0:
        # rooflines: one stream, every kernel launch stamped by a pair of HIP events (its own start and stop)
        nr = min(args.roofline_images, B)
        d1 = Detector(local, min(pass_size, nr), 1024, 1024)
        # first WITHOUT the per-launch event pairs: one event in front of and one behind the stage's launches on the stream, i.e. kernel
        # durations PLUS the dispatch gaps between them -- the quantity rounds 1-3 reported (the two methods differ by ~7 %;
        # both are in the line, `frac` is the kernel-duration one, the one a rocprofv3 kernel trace reproduces)
        d1.detect_device(dev[:nr], params)
        d1.detect_device(dev[:nr], params)
        stage_gaps_s = d1.last_timing()["blur_canny_ms"] * 1e-3
        d1.set_profiling(True)
        d1.detect_device(dev[:nr], params)
        d1.detect_device(dev[:nr], params)
        timing = d1.last_timing()
        seg = d1.last_kernel_timing()
        # votes of the HoughCircles stage on this workload: sum of the debug accumulators of a few of the same diagrams
        nv = min(4, nr)
        d1.set_debug(True)
        imgs = dev[:nv].cpu().numpy()
        d1.detect_batch(list(imgs), params, full=False)
        votes = sum(int(d1.fetch_circle_acc(i, v).sum()) for i in range(nv) for v in range(8)) / nv
        # the noisy variant of the same diagrams (sigma = 6): every median band falls back to the bit-serial kernel, hysteresis
        # has real work -- the headline fraction is the clean workload's, this is the honest companion
        gen = torch.Generator(device="cuda").manual_seed(1)
        noisy = (dev[:nr].float() + torch.randn(dev[:nr].shape, device="cuda", generator=gen) * NOISE_SIGMA).clamp(0, 255).to(torch.uint8)
        d1.set_debug(False)
        d1.detect_device(noisy, params)
        d1.detect_device(noisy, params)
        seg_noisy = d1.last_kernel_timing()
        del noisy
        arch, lib_path = d1.arch, d1.lib.path
        d1.close()
        stage_s = sum(seg.get(k, 0.0) for k in BLUR_CANNY_SEGS) * 1e-3
        stage_noisy_s = sum(seg_noisy.get(k, 0.0) for k in BLUR_CANNY_SEGS) * 1e-3
        canny7_s = sum(seg.get(k, 0.0) for k in CANNY7_SEGS) * 1e-3
        ach = BLUR_CANNY_BYTES * nr / stage_s / 1e9
        ach_noisy = BLUR_CANNY_BYTES * nr / stage_noisy_s / 1e9
        ach7 = CANNY7_BYTES * nr / canny7_s / 1e9
        traffic, traffic7, traffic_src = measured_traffic()
        vote_s = seg["k_vote_centres"] * 1e-3
        images = total * args.steps
        out = {
            "metric": "go-diagram images/sec (1024x1024 greyscale)", "value": images / dt, "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "device_arch": arch, "lib": os.path.relpath(lib_path, os.path.dirname(os.path.abspath(__file__))),
            "config": {"workload": "batch of %d synthetic 1024x1024 19x19 diagrams per GPU (BASELINE configs[%d]), "
                                   "device-resident, full hot path incl. board all-gather" % (B, 2 if world == 1 else 3),
                       "pass_size": pass_size, "streams": args.streams, "boards_match_generator": ok,
                       "generator_exceptions_rank0": exceptions,
                       "opencv_switches": dict(params.switch_set(), restates="OpenCV 4.3 .. 4.5.1 (package default, DESIGN.md 2a)"),
                       "rccl_ranks": world if use_dist else 0,
                       "allgather": "ncclAllGather of device-resident records through the C ABI" if use_dist else "single process: none",
                       "allgather_in_place": bool(gather.in_place()) if gather is not None else None},
            "roofline": {"bound": "hbm",
                         "kernel": "blur+Canny stage: " + ", ".join(k for k in BLUR_CANNY_SEGS if k in seg),
                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                         "frac_of_measured_copy_ceiling": ach / HBM_COPY_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_image": BLUR_CANNY_BYTES,
                         "stage_us_per_image": stage_s / nr * 1e6,
                         "stage_us_per_image_incl_dispatch_gaps": stage_gaps_s / nr * 1e6,
                         "frac_incl_dispatch_gaps": BLUR_CANNY_BYTES * nr / stage_gaps_s / 1e9 / HBM_PEAK_GBS,
                         "measured_on": "1 stream, %d diagrams, HIP start / stop events of every kernel launch on the context's stream "
                                        "(hipExtLaunchKernelGGL): kernel durations, as in a rocprofv3 kernel trace" % nr},
            "roofline_noisy": {"bound": "hbm", "kernel": "blur+Canny stage on the noisy variant of the workload (N(0, %g^2) added, clipped)" % NOISE_SIGMA,
                               "achieved": ach_noisy, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_noisy / HBM_PEAK_GBS,
                               "traffic": None, "algorithmic_bytes_per_image": BLUR_CANNY_BYTES,
                               "stage_us_per_image": stage_noisy_s / nr * 1e6,
                               "kernel_us_per_image": {k: seg_noisy[k] * 1e3 / nr for k in BLUR_CANNY_SEGS if k in seg_noisy}},
            "roofline_canny7": {"bound": "hbm", "kernel": "HoughCircles' internal Canny of 7 planes: " + ", ".join(CANNY7_SEGS),
                                "achieved": ach7, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach7 / HBM_PEAK_GBS,
                                "traffic": traffic7, "algorithmic_bytes_per_image": CANNY7_BYTES,
                                "us_per_image": canny7_s / nr * 1e6},
            "roofline_k5": {"bound": "lds-atomic", "kernel": "k_vote_centres (HoughCircles accumulator, 8 variants per image)",
                            "votes_per_image": votes, "achieved": votes * nr / vote_s, "peak": LDS_ATOMIC_PEAK, "unit": "votes/s",
                            "frac": votes * nr / vote_s / LDS_ATOMIC_PEAK,
                            "peak_source": "tools/micro/lds_atomic_bench.hip: 13.5 conflict-free atomic lanes per CU-cycle x 256 CUs x 2.4 GHz",
                            "us_per_image": vote_s / nr * 1e6,
                            "hough_circles_stage_us_per_image": timing["hough_circles_ms"] * 1e3 / nr},
            "kernel_us_per_image": {k: v * 1e3 / nr for k, v in seg.items()},
            "single_stream_stage_ms": timing,
        }
        if not args.no_fixtures:
            out["roofline_fixtures"] = fixtures_roofline(local)
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_per_worker)
        print(json.dumps(out))
        sys.stdout.flush()
    if gather is not None:
        gather.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
