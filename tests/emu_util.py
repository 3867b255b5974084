"""Loads the emulated build of the product sources (tests/emu) for GPU-less logic tests."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


def emu_library():
    """I2S_EMU_LIBRARY: an emulated build of an EXPERIMENT (tools/experiments/apply.py NAME --emu leaves build/exp/NAME/libi2s_emu.so), so
    that the whole emulated suite -- colour sources, fixtures, call sequences -- can be run on a patch before it is ever timed."""
    import build_emu
    from img2sgf_amd._lib import I2sLibrary
    override = os.environ.get("I2S_EMU_LIBRARY")
    if override:
        sys.stderr.write("tests: emulated kernels from %s\n" % override)
        return I2sLibrary(override)
    return I2sLibrary(build_emu.build())
