"""VERDICT r4 item 5a, counted before building: how many 256 x 32-pixel Canny bands (with their one-pixel apron) of the reference's fixtures,
after its contrast / brightness step (img2sgf.py:141-149), have equal channels everywhere -- and could take the single-channel walk instead of
the 3-plane mode -- and how many of those are two-valued (byte walk).  Uses the oracle's loader (test infrastructure): python tools/colour_band_stats.py"""
import sys, os, numpy as np
sys.path.insert(0,'.')
from oracle import pipeline as opipe
G='tests/golden/test_images'
names=["ex%d.jpg"%i for i in range(1,18)]+["no_circles.jpg"]
tot=[0,0,0,0,0]
for n in names:
    im=opipe.load_and_enhance(os.path.join(G,n))
    h,w,_=im.shape
    eq=(im[...,0]==im[...,1])&(im[...,1]==im[...,2])
    col = not eq.all()
    # bands: 256 wide x 64 tall (k_blur / canny bands are 256 x 32 rows for sobel?), use 256x32 for the Canny walk and 256x64 for blur
    def bands(bh):
        nb=0; nq=0; n2=0; nboth=0
        for y in range(0,h,bh):
            for x in range(0,w,256):
                # band reads +1 row/col apron
                y0,y1=max(y-1,0),min(y+bh+1,h); x0,x1=max(x-4,0),min(x+256+4,w)
                e=eq[y0:y1,x0:x1].all()
                sub=im[y0:y1,x0:x1,0]
                tv=((sub==0)|(sub==255)).all()
                nb+=1; nq+=e; n2+=tv; nboth+= (e and tv)
        return nb,nq,n2,nboth
    b32=bands(32)
    print("%-14s %4dx%4d colour=%d  bands(256x32): %4d  channel-equal %4d  two-valued(ch0) %4d  both %4d   eq px %.3f" % (n,w,h,col,*b32,eq.mean()))
    if col:
        for k in range(4): tot[k]+=b32[k]
        tot[4]+=1
print("colour fixtures:",tot)
