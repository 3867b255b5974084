"""Synthetic go-diagram generator for the benchmark and the parity tests.

Workload definition of BASELINE.json configs[1..3], `GEOM_1024`: 1024x1024 uint8, white
background, 19+19 black grid lines 2 px thick at x_k = 62 + 50 k (pixels x_k, x_k+1) spanning the grid extent
only, 9 hoshi discs of radius 3, every intersection independently empty / black / white with probability
0.55 / 0.225 / 0.225 from numpy Generator(PCG64(seed)); a black stone is a filled disc of radius 23 centred on
(x_k+0.5, y_k+0.5), a white stone a white disc with a 2-px black ring (radii 21..23) that hides the grid lines
under it.  Enters the pipeline where the reference holds `input_image_np` (img2sgf.py:150), C = 1.
Smaller geometries (same construction) keep the CPU-side parity tests fast.
"""
from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class Geometry:
    width: int
    height: int
    origin_x: int
    origin_y: int
    pitch: int
    nx: int          # grid lines across (columns)
    ny: int          # grid lines down (rows)
    r_out: int
    r_in: int
    hoshi_r: int = 3


GEOM_1024 = Geometry(1024, 1024, 62, 62, 50, 19, 19, 23, 21)
GEOM_SMALL = Geometry(272, 240, 24, 20, 26, 9, 8, 11, 9, 2)     # part board, fast to emulate
SIZE = 1024
N = 19


def _stamps(geom):
    r = geom.r_out + 1
    yy, xx = np.mgrid[-r:r + 2, -r:r + 2]          # pixel p covers offsets p - 0.5 from the centre (k+0.5)
    d2 = (2 * xx - 1) ** 2 + (2 * yy - 1) ** 2      # (2*distance)^2, exact integers
    return d2 <= (2 * geom.r_out) ** 2, d2 < (2 * geom.r_in) ** 2, r


def occupancy(seed, nx=19, ny=19):
    """(nx,ny) uint8 indexed [col, row] like the reference's full_board: 0 empty, 1 black, 2 white."""
    rng = np.random.Generator(np.random.PCG64(seed))
    u = rng.random((nx, ny))
    occ = np.zeros((nx, ny), np.uint8)
    occ[u >= 0.55] = 1
    occ[u >= 0.775] = 2
    return occ


def synth_diagram(seed, noisy=False, geom=GEOM_1024):
    """Returns (image HxW uint8, occupancy (nx,ny) uint8 [col,row])."""
    g = geom
    img = np.full((g.height, g.width), 255, np.uint8)
    x_lo, x_hi = g.origin_x, g.origin_x + g.pitch * (g.nx - 1) + 2
    y_lo, y_hi = g.origin_y, g.origin_y + g.pitch * (g.ny - 1) + 2
    for k in range(g.nx):
        c = g.origin_x + g.pitch * k
        img[y_lo:y_hi, c:c + 2] = 0
    for k in range(g.ny):
        c = g.origin_y + g.pitch * k
        img[c:c + 2, x_lo:x_hi] = 0
    hr = g.hoshi_r
    yy, xx = np.mgrid[-hr - 1:hr + 3, -hr - 1:hr + 3]
    hoshi = ((2 * xx - 1) ** 2 + (2 * yy - 1) ** 2) <= (2 * hr) ** 2
    if g.nx == 19 and g.ny == 19:
        pts = [(i, j) for i in (3, 9, 15) for j in (3, 9, 15)]
    else:
        pts = [(2, 2), (g.nx - 3, 2)] if g.nx >= 5 and g.ny >= 5 else []
    for i, j in pts:
        cx, cy = g.origin_x + g.pitch * i, g.origin_y + g.pitch * j
        sub = img[cy - hr - 1:cy + hr + 3, cx - hr - 1:cx + hr + 3]
        sub[hoshi] = 0
    occ = occupancy(seed, g.nx, g.ny)
    disc, inner, r = _stamps(g)
    for i in range(g.nx):
        for j in range(g.ny):
            o = occ[i, j]
            if not o:
                continue
            cx, cy = g.origin_x + g.pitch * i, g.origin_y + g.pitch * j
            ys, xs = max(cy - r, 0), max(cx - r, 0)
            sub = img[ys:cy + r + 2, xs:cx + r + 2]
            d = disc[ys - (cy - r):, xs - (cx - r):][:sub.shape[0], :sub.shape[1]]
            sub[d] = 0
            if o == 2:
                sub[inner[ys - (cy - r):, xs - (cx - r):][:sub.shape[0], :sub.shape[1]]] = 255
    if noisy:
        rng = np.random.Generator(np.random.PCG64(seed + (1 << 32)))
        img = np.clip(img.astype(np.float64) + rng.normal(0.0, 6.0, img.shape), 0, 255).astype(np.uint8)
    return img, occ


def synth_batch(seeds, noisy=False, geom=GEOM_1024):
    imgs = np.empty((len(seeds), geom.height, geom.width), np.uint8)
    occs = np.empty((len(seeds), geom.nx, geom.ny), np.uint8)
    for n, s in enumerate(seeds):
        imgs[n], occs[n] = synth_diagram(int(s), noisy, geom)
    return imgs, occs


# ---- batched rendering (torch, CPU or GPU) --------------------------------------------------------------------------
# In GEOM_1024 a stone (47 px) is narrower than the grid pitch (50 px), so the diagram is a 19 x 19 mosaic of 50 x 50
# cells whose content depends only on the cell position and its occupancy.  The cell library is cut out of three
# images rendered by synth_diagram() (all empty / all black / all white), so the mosaic is bit-identical to it.
_CELL_LIB = None


def _cell_library():
    global _CELL_LIB
    if _CELL_LIB is None:
        g = GEOM_1024
        lib = np.empty((3, N, N, g.pitch, g.pitch), np.uint8)          # [occ][col i][row j][y][x]
        for occ in range(3):
            img = _render(np.full((N, N), occ, np.uint8), g)
            for i in range(N):
                for j in range(N):
                    x0, y0 = g.origin_x + g.pitch * i - g.pitch // 2, g.origin_y + g.pitch * j - g.pitch // 2
                    lib[occ, i, j] = img[y0:y0 + g.pitch, x0:x0 + g.pitch]
        _CELL_LIB = lib
    return _CELL_LIB


def _render(occ, g):
    """synth_diagram's drawing with a given occupancy matrix."""
    saved = occupancy
    try:
        globals()["occupancy"] = lambda seed, nx=19, ny=19: occ
        return synth_diagram(0, False, g)[0]
    finally:
        globals()["occupancy"] = saved


def synth_batch_torch(seeds, device="cpu"):
    """(B,1024,1024) uint8 torch tensor on `device` and the (B,19,19) numpy occupancies; equals synth_batch(seeds)."""
    import torch
    g = GEOM_1024
    occs = np.stack([occupancy(int(s)) for s in seeds]) if len(seeds) else np.zeros((0, N, N), np.uint8)
    lib = torch.from_numpy(_cell_library()).to(device)                   # (3,19,19,50,50)
    o = torch.from_numpy(occs.astype(np.int64)).to(device)               # (B,19,19) [col i][row j]
    ii = torch.arange(N, device=device).view(1, N, 1).expand_as(o)
    jj = torch.arange(N, device=device).view(1, 1, N).expand_as(o)
    cells = lib[o, ii, jj]                                               # (B, i, j, y, x)
    mosaic = cells.permute(0, 2, 3, 1, 4).reshape(len(seeds), N * g.pitch, N * g.pitch)   # (B, j*50+y, i*50+x)
    out = torch.full((len(seeds), g.height, g.width), 255, dtype=torch.uint8, device=device)
    x0, y0 = g.origin_x - g.pitch // 2, g.origin_y - g.pitch // 2
    out[:, y0:y0 + N * g.pitch, x0:x0 + N * g.pitch] = mosaic
    return out, occs


# ---- what the reference's algorithm makes of a seed -------------------------------------------------------------------
_EXCEPTIONS = None


def algorithm_exceptions():
    """{seed: 19 x 19 board} for the seeds (searched: 0 .. 65535) whose board BY THE REFERENCE'S ALGORITHM is not the generator's
    occupancy -- e.g. seed 15634: 18 stones on the 19 points of the last column hide its grid line, HoughLines finds 18 vertical
    clusters, the board comes out 18 x 19.  Data written by tests/golden/make_synth_exceptions.py from the oracle's answers."""
    global _EXCEPTIONS
    if _EXCEPTIONS is None:
        import json
        import os
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_exceptions.json")) as f:
            _EXCEPTIONS = {int(s): np.array(e["board"], np.uint8) for s, e in json.load(f)["seeds"].items()}
    return _EXCEPTIONS


def exceptions_switch_set():
    """The OpenCV switch set (Params.switch_set() keys) the exception list was generated under."""
    import json
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_exceptions.json")) as f:
        return json.load(f)["opencv_switches"]


def expected_boards(seeds, occs, switches=None):
    """The boards a correct run of the hot path returns for `seeds`: the generator's occupancies `occs` (B, 19, 19), with the
    algorithm's own answer substituted for the few seeds of algorithm_exceptions().  Returns (boards, exception seeds in range).
    switches: the run's Params.switch_set(); the list was searched and confirmed under ONE switch set and is refused for any other
    (another angle count or tap set may read other seeds differently)."""
    if switches is not None and dict(switches) != exceptions_switch_set():
        raise ValueError("synth_exceptions.json was generated under %s, the run uses %s: regenerate it "
                         "(tools/synth_mismatches.py, tests/golden/make_synth_exceptions.py)" % (exceptions_switch_set(), dict(switches)))
    exc = algorithm_exceptions()
    want = np.array(occs, np.uint8, copy=True)
    hit = []
    for k, s in enumerate(seeds):
        if int(s) in exc:
            want[k] = exc[int(s)]
            hit.append(int(s))
    return want, hit
