"""Canny row kernels at every column-group boundary (a wavefront walks 256 columns; lanes 0 / 63 need the magnitude of the column beyond):
random 8-bit, two-valued and colour images of widths 253 .. 1028 x heights 30 .. 67, every Canny map, plane and record against the oracle, with
the fused and the unfused threshold pairs -- on an EMULATED build given by path (the product's: tests/emu/libi2s_emu.so; an experiment's:
build/exp/NAME/libi2s_emu.so after `tools/experiments/apply.py NAME --emu`).  Written for canny_lean.patch (round 6), whose end-column
magnitudes are computed once per band.      python tests/stress/canny_widths.py PATH/libi2s_emu.so       (~6 min)"""
import os
import sys

import numpy as np

TESTS = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(TESTS))
sys.path.insert(0, TESTS)
import parity
from img2sgf_amd._lib import I2sLibrary
from img2sgf_amd.pipeline import Detector, Params
lib=I2sLibrary(sys.argv[1])
rng=np.random.default_rng(42)
n=0
for w in (253,255,256,257,259,260,261,511,512,513,516,1021,1024,1025,1028):
    for h in (30,33,64,67):
        det=Detector(0,3,w,h,lib=lib)
        g=rng.integers(0,256,(h,w),dtype=np.uint8)
        two=(rng.integers(0,2,(h,w))*255).astype(np.uint8)
        col=rng.integers(0,256,(h,w,3),dtype=np.uint8)
        parity.run_and_compare(det,[g,two,col],internals=True)
        # a main-Canny threshold pair different from HoughCircles' (unfused modes 1 and 0 on the grey plane)
        parity.run_and_compare(det,[g,two,col],params=Params(canny_lo=40,canny_hi=150),oracle_kwargs=dict(canny=(40,150)))
        det.close(); n+=6
print("canny widths x heights: %d images exact (%s)"%(n,sys.argv[1]))
