#!/usr/bin/env python3
"""Per-kernel HIP-event times of the hot path on the benchmark workload, one stream (quick A/B tool for kernel work).
usage: tools/kernel_times.py [--lib path/to/libi2s_hip.so] [--images 128] [--pass-size 128] [--reps 3]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--images", type=int, default=128)
    ap.add_argument("--pass-size", type=int, default=128)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--noisy", action="store_true")
    args = ap.parse_args()
    import torch
    from img2sgf_amd import _lib, synth
    from img2sgf_amd.pipeline import Detector, Params
    lib = _lib.I2sLibrary(args.lib) if args.lib else None
    dev, occs = synth.synth_batch_torch(range(args.images), torch.device("cuda", 0))
    if args.noisy:
        g = torch.Generator(device="cuda").manual_seed(1)
        dev = (dev.float() + torch.randn(dev.shape, device="cuda", generator=g) * 6).clamp(0, 255).to(torch.uint8)
    det = Detector(0, args.pass_size, 1024, 1024, lib=lib)
    det.set_profiling(True)
    best = None
    for _ in range(args.reps):
        boards = det.detect_device(dev, Params())
        seg = det.last_kernel_timing()
        if best is None or sum(seg.values()) < sum(best.values()):
            best = seg
    ok = all((__import__("numpy").ctypeslib.as_array(boards[k].board) == occs[k]).all() for k in range(args.images)) if not args.noisy else None
    per = {k: round(v * 1e3 / args.images, 3) for k, v in best.items()}
    per["TOTAL"] = round(sum(best.values()) * 1e3 / args.images, 3)
    print(json.dumps({"lib": args.lib or "default", "images": args.images, "boards_ok": ok, "us_per_image": per}))


if __name__ == "__main__":
    main()
