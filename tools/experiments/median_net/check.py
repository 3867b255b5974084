"""The prototype kernel (median_net.hip) on the fiber emulation of tests/emu against numpy's 5x5 / 7x7 medians with replicated borders
(cv.medianBlur's border rule) -- random, two-valued and tie-heavy images, sizes that exercise both column strips and the image edges.
usage: python tools/experiments/median_net/check.py"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
EMU = os.path.join(ROOT, "tests", "emu")


def median(img, K):
    h = K // 2
    p = np.pad(img, h, mode="edge")
    win = np.lib.stride_tricks.sliding_window_view(p, (K, K)).reshape(img.shape + (K * K,))
    return np.sort(win, axis=-1)[..., (K * K - 1) // 2].astype(np.uint8)


def main():
    subprocess.check_call([sys.executable, os.path.join(HERE, "gen.py")], stdout=subprocess.DEVNULL)
    exe = "/tmp/median_net_emu"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", EMU, "-x", "c++", os.path.join(HERE, "median_net.hip"), os.path.join(EMU, "hipemu.cpp"), "-o", exe,
                           "-Wno-unknown-pragmas", "-lpthread"])
    rng = np.random.default_rng(3)
    cases = [(8, 4, "random"), (500, 37, "random"), (496, 18, "two"), (1024, 33, "random"), (300, 20, "ties"), (252, 9, "random"), (4, 1, "random")]
    for w, h, kind in cases:
        img = {"random": lambda: rng.integers(0, 256, (h, w)), "two": lambda: rng.integers(0, 2, (h, w)) * 255, "ties": lambda: rng.integers(100, 104, (h, w))}[kind]().astype(np.uint8)
        out = subprocess.run([exe], input=np.array([w, h], np.int32).tobytes() + img.tobytes(), capture_output=True, check=True).stdout
        got = np.frombuffer(out, np.uint8).reshape(2, h, w)
        ok5, ok7 = (got[0] == median(img, 5)).all(), (got[1] == median(img, 7)).all()
        print("%4d x %-3d %-6s  5x5 %s  7x7 %s" % (w, h, kind, "exact" if ok5 else "MISMATCH", "exact" if ok7 else "MISMATCH"))
        assert ok5 and ok7


if __name__ == "__main__":
    main()
