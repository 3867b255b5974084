"""Differential fuzzing of ONE long-lived context: the other fuzz tests make a fresh Detector per seed, so nothing there would
notice state that leaks from one call into the next (band flags, worklists and their stamps, the adaptive number of hysteresis
launches, counters, staging and record buffers sized on first use).  Per seed one context serves a random sequence of calls --
batches of 1 .. 9 ragged images (one to three device passes, sometimes area-scheduled), ordinary / extreme / switch-set parameters,
board records only or full records, the debug accumulators and the per-kernel profiling switched on and off, the occasional
re-classification -- and every call is checked against the oracle like a first call would be."""
import os

import numpy as np
import pytest

import parity
from test_gpu_fuzz import _random_image, _random_params
from test_gpu_fuzz_extreme import _extreme_params
from img2sgf_amd.pipeline import Detector, board_to_sgf
from oracle import pipeline as opipe

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("I2S_FUZZ_SEQ_SEEDS", 12))


def run_call_sequence(det, rng, tag, n_calls=10, side=330, max_images=9):
    """The sequence itself (also driven by the emulated twin in test_emu_pipeline.py, on small images)."""
    for call in range(n_calls):
        imgs = [np.ascontiguousarray(_random_image(rng)[:side, :side]) for _ in range(int(rng.integers(1, max_images + 1)))]
        params, okw = _extreme_params(rng) if rng.random() < 0.2 else _random_params(rng, int(rng.integers(0, 1000)))
        params.schedule = bool(rng.random() < 0.4)
        det.set_profiling(bool(rng.random() < 0.3))
        internals = set(okw) <= {"compat"} and rng.random() < 0.4
        if not internals:
            det.set_debug(False)
        boards = det.detect_batch(imgs, params, full=False)                 # board records only first: the same call must follow it
        keep = [k for k in range(len(imgs)) if boards[k].status != 100]
        if rng.random() < 0.5 and keep:
            params.schedule = False                     # (run_and_compare reads the planes of the LAST pass in input order)
            dets = parity.run_and_compare(det, [imgs[k] for k in keep], params=params, internals=internals, oracle_kwargs=okw)
            for k, d in zip(keep, dets):
                assert d.board_ready == (boards[k].status == 0), (tag, call, k)
            if rng.random() < 0.5:
                # apply_black_thresh (img2sgf.py:762-766): identify_board alone on the images of the last device pass, twice
                kept = [imgs[k] for k in keep]
                nb_last = (len(kept) - 1) % det.max_batch + 1
                for _ in range(2):
                    thr, al = int(rng.integers(0, 256)), (2 + int(rng.integers(0, 2)), int(rng.integers(0, 2)))
                    params.black_threshold, params.alignment = thr, al
                    again = det.classify(0, nb_last, params)
                    for q, d in enumerate(again):
                        parity.compare_detection(d, opipe.process_image(kept[len(kept) - nb_last + q], **dict(okw, black_thr=thr, alignment=al)))
        else:
            for k in keep:
                ref = opipe.process_image(imgs[k], **okw)
                assert (board_to_sgf(boards[k]) if boards[k].status == 0 else None) == ref["sgf"], (tag, call, k)


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_call_sequences_on_one_context(seed):
    rng = np.random.default_rng(70000 + seed)
    det = Detector(0, int(rng.integers(1, 5)), 330, 330)
    run_call_sequence(det, rng, seed)
    det.close()
