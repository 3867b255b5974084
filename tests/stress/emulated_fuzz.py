"""The GPU suite's differential fuzz families on the EMULATED kernels (tests/emu: the product's kernel and host sources compiled
against the fiber-based HIP emulation, csrc/isa/gfx950_ops.h replaced by plain C), many seeds, several processes -- for rounds without a
GPU: it checks the kernels' LOGIC at HEAD against the oracle, not the machine-level paths.

    python tests/stress/emulated_fuzz.py [--family fuzz|extreme|sequences|medium|preprocess|jpeg|fixtures] [--first 0] [--seeds 200] [--procs 8]
"""
import argparse
import os
import sys
import time
import traceback
from concurrent.futures import ProcessPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, os.path.dirname(TESTS))


def one(args):
    family, seed = args
    import numpy as np
    import emu_util
    from img2sgf_amd.pipeline import Detector
    lib = emu_util.emu_library()
    mk = lambda nb, w, h: Detector(0, nb, w, h, lib=lib)
    try:
        if family == "fuzz":
            from test_gpu_fuzz import run_fuzz_seed
            run_fuzz_seed(mk, seed)
        elif family == "preprocess":
            from test_gpu_fuzz import run_preprocessing_fuzz_seed
            run_preprocessing_fuzz_seed(mk, seed)
        elif family == "extreme":
            from test_gpu_fuzz_extreme import run_extreme_seed
            run_extreme_seed(mk, seed)
        elif family == "medium":
            from test_gpu_fuzz_medium import run_medium_seed
            run_medium_seed(mk, seed, n_images=1)
        elif family == "sequences":
            from test_gpu_fuzz_sequences import run_call_sequence
            rng = np.random.default_rng(70000 + seed)
            det = mk(int(rng.integers(1, 5)), 330, 330)
            run_call_sequence(det, rng, seed)
            det.close()
        elif family == "jpeg":
            from test_gpu_fuzz_jpeg_sequences import run_jpeg_call_sequence
            rng = np.random.default_rng(210000 + seed)
            nb = int(rng.integers(1, 5))
            det, ref = mk(nb, 310, 310), mk(nb, 310, 310)
            run_jpeg_call_sequence(det, ref, rng, seed)
            det.close(); ref.close()
        elif family == "fixtures":
            # the reference's 18 scans x the six OpenCV switch sets (108 "seeds"): every plane, accumulator, list and record
            import parity
            import switches
            from helpers import GOLDEN
            from oracle import pipeline as opipe
            names = ["ex%d.jpg" % i for i in range(1, 18)] + ["no_circles.jpg"]
            name, sw = names[seed % 18], switches.NAMES[(seed // 18) % len(switches.NAMES)]
            img = opipe.load_and_enhance(os.path.join(GOLDEN, "test_images", name))
            det = mk(1, img.shape[1], img.shape[0])
            parity.run_and_compare(det, [img], params=switches.params(sw), internals=seed < 36, oracle_kwargs=dict(compat=switches.compat(sw)))
            det.close()
        else:
            raise ValueError(family)
        return seed, None
    except BaseException:
        return seed, traceback.format_exc()


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--family", default="fuzz")
    ap.add_argument("--first", type=int, default=0)
    ap.add_argument("--seeds", type=int, default=200)
    ap.add_argument("--procs", type=int, default=8)
    a = ap.parse_args()
    import emu_util
    emu_util.emu_library()                       # build once, before the workers start
    t0 = time.time()
    bad = []
    with ProcessPoolExecutor(a.procs) as ex:
        for seed, err in ex.map(one, [(a.family, s) for s in range(a.first, a.first + a.seeds)]):
            if err:
                bad.append(seed)
                print("seed %d FAILED\n%s" % (seed, err), flush=True)
    print("%s: seeds %d .. %d on the emulated kernels, %d processes: %d failed %s in %.0f s" % (
        a.family, a.first, a.first + a.seeds - 1, a.procs, len(bad), bad, time.time() - t0))
    sys.exit(1 if bad else 0)
