"""Builds tests/emu/libi2s_emu.so: the product's kernel and host sources (img2sgf_amd/csrc/*) compiled with g++ against
the fiber-based HIP emulation in tests/emu/hip/hip_runtime.h.  ONE product header is replaced: csrc/isa/gfx950_ops.h (inline
assembly, DPP, buffer descriptors) by tests/emu/gfx950_ops.h (plain C with the same semantics); those machine-level paths are
covered by the GPU tests only.  Test infrastructure only: lets the GPU-less CI
check kernel logic against the oracle.  The product loader never looks at this file."""
import fcntl
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "img2sgf_amd", "csrc")
LIB = os.path.join(HERE, "libi2s_emu.so")


def build(force=False, csrc=None, out=None):
    """csrc / out: an alternative source directory and library path -- the experiments of tools/experiments/ are checked for
    bit-exactness on the emulated kernels before they are ever timed (tools/experiments/apply.py)."""
    global CSRC, LIB
    if csrc is not None:
        saved = CSRC, LIB
        CSRC, LIB = csrc, out
        try:
            return build(force)
        finally:
            CSRC, LIB = saved
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not os.path.isdir(os.path.join(CSRC, f))] + [
        os.path.join(HERE, "hipemu.cpp"), os.path.join(HERE, "hip", "hip_runtime.h"), os.path.join(HERE, "gfx950_ops.h"),
        os.path.join(ROOT, "include", "i2s.h")]
    def fresh():
        return os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps)
    if not force and fresh():
        return LIB
    # pytest-xdist workers arrive here together: one builds (into a temporary name, renamed when complete), the others wait
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if force or not fresh():
            tmp = LIB + ".%d.tmp" % os.getpid()
            cmd = ["g++", "-O2", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
                   "-Wall", "-Wno-unused-function", "-Wno-unknown-pragmas", "-Wno-sign-compare",
                   "-I", HERE, "-x", "c++", os.path.join(CSRC, "i2s_api.hip"), os.path.join(HERE, "hipemu.cpp"),
                   "-o", tmp]
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
