"""One long-lived context decoding random sequences of JPEG batches (the per-seed contexts of test_gpu_jpeg.py would not notice state
that leaks between calls: the blob / coefficient / table buffers grown on demand, the subsequence stamps, the round flags, the files a
pass hands back to the serial decoder): 1 .. 9 Pillow-encoded files per call (every encoder setting, 1 .. 3 device passes), a random
entropy path per call (host threads / parallel on the device / one lane per file), sometimes area-scheduled.  The records must equal
those of the array path on Pillow's own decode, and the staged sources of the last pass Pillow's pixels."""
import os

import numpy as np
import pytest

from test_gpu_jpeg import _encode_random
from img2sgf_amd.pipeline import Detector, Params

pytestmark = pytest.mark.gpu

N_SEEDS = int(os.environ.get("I2S_FUZZ_JPEG_SEQ_SEEDS", 6))


def run_jpeg_call_sequence(det, ref, rng, tag, n_calls=8, max_files=9):
    """The sequence itself (also driven by the emulated twin in test_preflight_gpu_suite.py, shorter)."""
    nb = det.max_batch
    for call in range(n_calls):
        pairs = [_encode_random(rng) for _ in range(int(rng.integers(1, max_files + 1)))]
        blobs, pix = [p[0] for p in pairs], [p[1] for p in pairs]
        sched = bool(rng.random() < 0.3)
        p = Params(jpeg_entropy_device=int(rng.integers(0, 3)), schedule=sched)
        got = det.detect_jpeg(blobs, p, full=False)
        want = ref.detect_batch(pix, Params(schedule=sched), full=False)
        for k in range(len(blobs)):
            assert bytes(got[k]) == bytes(want[k]), (tag, call, k)
        if not sched:
            n_last = (len(blobs) - 1) % nb + 1
            for q in range(n_last):
                np.testing.assert_array_equal(det.fetch_source(q, 3), pix[len(blobs) - n_last + q], err_msg="seed %s call %d image %d" % (tag, call, q))


@pytest.mark.parametrize("seed", range(N_SEEDS))
def test_fuzz_jpeg_call_sequences_on_one_context(seed):
    rng = np.random.default_rng(210000 + seed)
    nb = int(rng.integers(1, 5))
    det = Detector(0, nb, 310, 310)
    ref = Detector(0, nb, 310, 310)
    run_jpeg_call_sequence(det, ref, rng, seed)
    det.close(); ref.close()
