"""Device-resident inputs with arbitrary geometry: odd widths, padded strides, base pointers off by 0..3 bytes, 1 and 3 channels,
mixed in one pass; against the host path's records, and the planes against the oracle for some."""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")   # run from the repo root
import numpy as np, torch
from img2sgf_amd.pipeline import Detector, Params
from test_gpu_fuzz import _random_image
import parity
from oracle import pipeline as opipe
rng = np.random.default_rng(11)
det_d = Detector(0, 4, 340, 340); det_h = Detector(0, 4, 340, 340)
bad = 0
for trial in range(150):
    imgs = [np.ascontiguousarray(_random_image(rng)[:330, :330]) for _ in range(int(rng.integers(1, 7)))]
    bufs, ptrs, strides = [], [], []
    for im in imgs:
        h, w = im.shape[:2]; cn = 1 if im.ndim == 2 else 3
        pad = int(rng.choice([0, 0, 1, 3, 4, 13])); off = int(rng.choice([0, 0, 1, 2, 3]))
        stride = w * cn + pad
        host = np.zeros(off + stride * h + 8, np.uint8)
        view = np.lib.stride_tricks.as_strided(host[off:], (h, w * cn), (stride, 1))
        view[:] = im.reshape(h, w * cn)
        t = torch.from_numpy(host).cuda()
        bufs.append(t); ptrs.append(t.data_ptr() + off); strides.append(stride)
    args = ([im.shape[1] for im in imgs], [im.shape[0] for im in imgs])
    chans = [1 if im.ndim == 2 else 3 for im in imgs]
    p = Params(contrast=int(rng.integers(0, 101)), brightness=int(rng.integers(0, 101))) if rng.random() < 0.25 else Params()
    bd, _ = det_d.detect_ptrs(ptrs, args[0], args[1], strides, chans, p, True)
    bh, _ = det_h.detect_ptrs([im.ctypes.data for im in imgs], args[0], args[1], [im.strides[0] for im in imgs], chans, p, False)
    for k in range(len(imgs)):
        if bytes(bd[k]) != bytes(bh[k]):
            bad += 1; print("MISMATCH trial", trial, "image", k, imgs[k].shape, strides[k], bd[k].status, bh[k].status)
    nb_last = (len(imgs) - 1) % 4 + 1
    for k in range(nb_last):
        for plane in ("grey", "edges", "median5", "gauss7", "removed"):
            if not np.array_equal(det_d.fetch_plane(k, plane), det_h.fetch_plane(k, plane)):
                bad += 1; print("PLANE MISMATCH trial", trial, k, plane)
    torch.cuda.synchronize()
print("device-input geometry trials: 150, mismatches", bad)
