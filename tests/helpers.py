"""Shared test helpers (data generators that must be identical on both sides of a parity check)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def synth_grey(w, h, seed):
    """Same integer-only texture as tests/golden/make_glue_golden.py:synth_grey."""
    y, x = np.mgrid[0:h, 0:w].astype(np.int64)
    a, b, c = 7 + seed % 13, 11 + seed % 17, 23 + seed % 29
    v = (x * a + y * b + ((x * y) % c) * 5 + ((x // 9 + y // 7 + seed) % 2) * 120) % 256
    return v.astype(np.uint8)


def load_glue_golden():
    with open(os.path.join(GOLDEN, "glue_golden.json")) as f:
        return json.load(f)


def free_port():
    """A TCP port on 127.0.0.1 that is free now (for a torch.distributed rendez-vous of a test: no fixed numbers, two suites on one
    box -- or a socket of an earlier run still in TIME_WAIT -- must not collide)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]
