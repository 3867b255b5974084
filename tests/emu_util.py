"""Loads the emulated build of the product sources (tests/emu) for GPU-less logic tests."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))


def emu_library():
    import build_emu
    from img2sgf_amd._lib import I2sLibrary
    return I2sLibrary(build_emu.build())
