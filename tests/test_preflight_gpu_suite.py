"""Pre-flight of the suite the driver runs with `pytest -m gpu -x`: GPU-only test code never executes in the
build container, and with -x one slip in it hides every test collected after it.  Three layers, all on the CPU:

* static (tests/static_check.py): every tests/test_gpu_*.py and the helper modules they share -- undefined names, attributes of
  imported modules that do not exist, argument lists that do not bind, methods a Detector / StreamedDetector / BoardGather does not
  have; the checker itself is checked against a module of seeded faults;
* collection: `pytest --collect-only -m gpu` in a subprocess collects without error and finds at least the 403 tests round 4 had;
* execution at toy size on the emulated kernels (tests/emu) of the bodies that have no emulated twin elsewhere: test_gpu_full_size's
  `_check`, `run_config2`, `run_config3` (small diagrams in a tensor stand-in, a one-rank stand-in communicator) and the JPEG
  call-sequence fuzz (the other shared bodies -- tests/parity.py, run_call_sequence, the extreme-parameter fuzz -- are driven by
  tests/test_emu_pipeline.py).

Expected wall time of the GPU suite on one MI355X box: 131 s for the 338 tests of round 4's last full run
(profiles/r04_g_final_runs.txt) + about 40 s for the 65 tests added since (fuzz seeds at their default counts) -- under 4 minutes
against the driver's limit of 1 200 s."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest

import emu_util
import static_check
from img2sgf_amd import dist as i2s_dist, synth

HERE = os.path.dirname(os.path.abspath(__file__))
GPU_MODULES = sorted(glob.glob(os.path.join(HERE, "test_gpu_*.py")))
SHARED = [os.path.join(HERE, f) for f in ("parity.py", "helpers.py", "switches.py", "jpeg_transcode.py", "_dist_worker.py", "test_dist_gloo.py")]
# the product's host layer and the bench: paths that only run on a GPU box (detect_device, BoardGather, `--opencv auto`, the roofline legs)
PRODUCT = [os.path.join(os.path.dirname(HERE), "img2sgf_amd", f) for f in ("pipeline.py", "dist.py", "gui_adapter.py", "__main__.py", "_lib.py", "synth.py",
                                                                            "preprocess.py")] + [os.path.join(os.path.dirname(HERE), "bench.py")]


@pytest.mark.parametrize("path", GPU_MODULES + SHARED + PRODUCT, ids=os.path.basename)
def test_static_check(path):
    bad = static_check.check_module(path)
    assert not bad, "\n".join(bad)


def test_static_checker_catches_seeded_faults(tmp_path, monkeypatch):
    src = '''
import parity
from img2sgf_amd import synth
from img2sgf_amd.pipeline import Detector, Params

def helper(a, b=1):
    return a

def test_a():
    det = Detector(0, 2, 300, 260)
    x = undefined_thing + 1
    parity.no_such_function(det)
    parity.compare_detection(det)
    helper(1, 2, 3)
    det.detect_batch([], Params(), bogus=1)
    det.no_method()
    synth.synth_diagram(1, geom=synth.GEOM_TINY)
    Detector(0, 1, 2, 3, 4, 5, 6)
    Params(not_a_field=3)
'''
    p = tmp_path / "seeded_faults_module.py"
    p.write_text(src)
    monkeypatch.syspath_prepend(str(tmp_path))
    bad = "\\n".join(static_check.check_module(str(p)))
    for needle in ("undefined_thing", "no_such_function", "missing a required argument: 'ref'", "call of helper: too many positional",
                   "unexpected keyword argument 'bogus'", "no method 'no_method'", "GEOM_TINY", "call of Detector: too many positional",
                   "unexpected keyword argument 'not_a_field'"):
        assert needle in bad, (needle, bad)


def test_gpu_suite_collects():
    out = subprocess.run([sys.executable, "-m", "pytest", HERE, "-m", "gpu", "--collect-only", "-q", "-p", "no:cacheprovider"],
                         capture_output=True, text=True, cwd=os.path.dirname(HERE), timeout=600)
    tail = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-500:]
    assert out.returncode == 0 and "error" not in tail.lower(), out.stdout[-2000:] + out.stderr[-2000:]
    n = int(tail.split("/")[0].split()[0])                 # "412/590 tests collected (178 deselected) in 3.2s"
    assert "deselected" in tail, tail
    assert n >= 403, tail


# ---- execution at toy size on the emulated kernels -------------------------------------------------------------------

class _DeviceTensorStandIn:
    """What Detector.detect_device needs of a CUDA tensor, over a numpy array (the emulated library's "device" pointers are host
    pointers)."""

    class _U8:
        def __str__(self):
            return "torch.uint8"

    def __init__(self, a):
        self.a = np.ascontiguousarray(a, np.uint8)
        self.is_cuda, self.dtype, self.shape = True, self._U8(), self.a.shape

    def is_contiguous(self):
        return True

    def dim(self):
        return self.a.ndim

    def data_ptr(self):
        return self.a.ctypes.data

    def __getitem__(self, k):
        return _DeviceTensorStandIn(self.a[k])


def _small_batch(seeds):
    """Small part-board diagrams; the occupancies padded to the 19 x 19 record (alignment left / top: the board's origin)."""
    imgs, occs = synth.synth_batch(list(seeds), geom=synth.GEOM_SMALL)
    full = np.zeros((len(occs), 19, 19), np.uint8)
    full[:, :occs.shape[1], :occs.shape[2]] = occs
    return _DeviceTensorStandIn(imgs), full


def test_full_size_check_helper_passes_and_fails():
    import test_gpu_full_size as fs
    seeds = range(15630, 15640)                               # holds 15634, the one exception below 32768
    occs = np.stack([synth.occupancy(s) for s in seeds])
    want, hit = synth.expected_boards(seeds, occs)
    assert hit == [15634]
    table = np.zeros((len(occs), 384), np.uint8)
    table[:, :361] = want.reshape(len(occs), 361)
    fs._check(table, occs, 15630, [15634])
    with pytest.raises(AssertionError):
        fs._check(table, occs, 15630, [])                     # the exception must be announced
    table[7, 5] ^= 1
    with pytest.raises(AssertionError, match="first seed 15637"):
        fs._check(table, occs, 15630, [15634])


class _OneRankCommLib:
    """The emulated library with the i2s_comm_* entry points answered by test_dist_gloo's stand-in communicator (librccl is not
    loadable without a GPU): one rank, its gather buffer in host memory = the emulated library's device memory."""

    def __init__(self, emu, total):
        from test_dist_gloo import _FakeCommDll
        fake = _FakeCommDll(1, i2s_dist.records_per_rank(total, 1))

        class Dll:
            def __getattr__(self, name):
                if name == "i2s_comm_unique_id":
                    return lambda idb: 0
                if name.startswith("i2s_comm_") or name == "i2s_allgather_boards":
                    return getattr(fake, name)
                return getattr(emu.dll, name)
        self.dll, self.path = Dll(), emu.path


def test_full_size_bodies_on_emulated_kernels():
    import test_gpu_full_size as fs
    emu = emu_util.emu_library()
    # one stream: the emulation is single-threaded
    fs.run_config2(3, make_batch=_small_batch, streams=1, pass_size=2, side=300, lib=emu, size=(9, 8))
    total, world = 5, 4                                       # shards 2, 1, 1, 1
    fs.run_config3(total, world, [], make_batch=_small_batch, streams=1, pass_size=2, side=300, lib=_OneRankCommLib(emu, total))


def test_jpeg_call_sequence_on_emulated_kernels():
    import test_gpu_fuzz_jpeg_sequences as js
    from img2sgf_amd.pipeline import Detector
    emu = emu_util.emu_library()
    rng = np.random.default_rng(210000)
    det, ref = Detector(0, 2, 310, 310, lib=emu), Detector(0, 2, 310, 310, lib=emu)
    js.run_jpeg_call_sequence(det, ref, rng, "emu", n_calls=2, max_files=3)
    det.close(); ref.close()


def test_gpu_suite_on_a_cpu_library_does_not_read_as_a_gpu_run(tmp_path):
    """with I2S_LIBRARY naming the CPU emulation under the product's file name, `pytest -m gpu -k native_library`
    FAILS (the context's device answers "emulated", i2s_device_arch), with and without the I2S_EXPERIMENT=1 declaration, and the
    loader says on stderr which file it runs on.  Nothing named libi2s_hip.so can make the GPU suite pass on a CPU."""
    import shutil
    import subprocess
    ROOT = os.path.dirname(HERE)
    fake = tmp_path / "libi2s_hip.so"
    shutil.copy(emu_util.emu_library().path, fake)
    for declared in ("0", "1"):
        env = dict(os.environ, I2S_LIBRARY=str(fake), I2S_EXPERIMENT=declared)
        out = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-q", "-x", "-k",
                              "native_library", "-p", "no:cacheprovider"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        assert out.returncode != 0 and "1 failed" in out.stdout, out.stdout[-800:] + out.stderr[-400:]
        assert ("I2S_LIBRARY redirects the suite" in out.stdout) if declared == "0" else ("'emulated', not on an MI355X" in out.stdout), out.stdout[-1200:]
        assert "I2S_LIBRARY is set -- running on " + str(fake) in out.stdout + out.stderr
