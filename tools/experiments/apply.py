"""Kernel experiments kept OUT of the product: each is a patch against img2sgf_amd/csrc (tools/experiments/<name>.patch).  This round
(no GPU) an experiment can be proven bit-exact on the emulated kernels here and timed later:

    python tools/experiments/apply.py NAME --emu     # copy csrc + patch -> build/exp/NAME/, build the emulated library, run the parity families on it
    python tools/experiments/apply.py NAME --hip     # ... and hipcc -> build/exp/NAME/libi2s_hip.so   (then: tools/kernel_times.py --lib that)
    tools/experiments/ab.sh NAME [NAME ...]          # on a GPU box: product vs experiments, kernel times, three runs each

An experiment that measures better and passes the GPU suite is merged into csrc; one that does not is logged in profiles/ and its
patch stays here as the record of what was tried."""
import argparse
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))


def tree(name):
    """build/exp/NAME/pkg/csrc (+ build/exp/NAME/include: the sources include ../../include/i2s.h) with the patch applied."""
    base = os.path.join(ROOT, "build", "exp", name)
    csrc = os.path.join(base, "pkg", "csrc")
    shutil.rmtree(base, ignore_errors=True)
    shutil.copytree(os.path.join(ROOT, "img2sgf_amd", "csrc"), csrc)
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(base, "include"))
    parts = name.split("+")
    for i, part in enumerate(parts):                                              # "tile128+cull_fast": several patches, in this order
        path = os.path.join(HERE, part + ".patch")
        if not os.path.exists(path):
            sys.exit("no such experiment: %s (tools/experiments/%s.patch)" % (part, part))
        with open(path) as f:                                                     # first line of a patch may say "# requires: tile128"
            first = f.readline()
        need = first.split(":", 1)[1].split() if first.startswith("# requires:") else []
        missing = [n for n in need if n not in parts[:i]]
        if missing:
            sys.exit("%s.patch applies on top of %s only: use  apply.py %s" % (part, " + ".join(need), "+".join(need + [part])))
        if subprocess.call(["patch", "-p3", "-s", "-d", csrc, "-i", path]) != 0:      # paths in the patch: img2sgf_amd/csrc/<file>
            sys.exit("%s.patch does not apply to the current img2sgf_amd/csrc%s" % (part, " + " + "+".join(parts[:i]) if i else ""))
    return base, csrc


def build_hip(name, csrc, base):
    from img2sgf_amd import build
    out = os.path.join(base, "libi2s_hip.so")
    flags = [f if f != os.path.join(build.CSRC, "isa") else os.path.join(csrc, "isa") for f in build.FLAGS]
    subprocess.check_call(["hipcc"] + flags + ["-o", out, os.path.join(csrc, "i2s_api.hip")])
    return out


def emulated_parity(name, csrc, base, seeds):
    import build_emu
    from img2sgf_amd._lib import I2sLibrary
    from img2sgf_amd.pipeline import Detector
    lib = I2sLibrary(build_emu.build(csrc=csrc, out=os.path.join(base, "libi2s_emu.so")))
    mk = lambda nb, w, h: Detector(0, nb, w, h, lib=lib)
    import parity
    from img2sgf_amd import synth
    from test_gpu_fuzz import run_fuzz_seed
    from test_gpu_fuzz_extreme import run_extreme_seed
    det = mk(2, 1024, 1024)
    parity.run_and_compare(det, [synth.synth_diagram(s)[0] for s in (0, 1)], internals=True)      # full-size diagrams, accumulators compared
    parity.run_and_compare(det, [synth.synth_diagram(2, noisy=True)[0]], internals=True)
    det.close()
    for s in range(seeds):
        run_fuzz_seed(mk, s)
        run_extreme_seed(mk, s)
    print("%s: bit-exact on the emulated kernels (3 full-size diagrams with accumulators, %d + %d fuzz seeds)" % (name, seeds, seeds))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name")
    ap.add_argument("--emu", action="store_true")
    ap.add_argument("--hip", action="store_true")
    ap.add_argument("--seeds", type=int, default=40)
    a = ap.parse_args()
    base, csrc = tree(a.name)
    if a.emu:
        emulated_parity(a.name, csrc, base, a.seeds)
    if a.hip:
        print(build_hip(a.name, csrc, base))
