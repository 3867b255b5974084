"""LDS bank behaviour of k_vote_centres' walk under different row strides of the accumulator tile -- a CPU model (no GPU this round), on
top of tools/vote_cull_model.py's reconstruction of the edge records.  The items of a tile are formed as the kernel forms them (bin by
bin, 64 records per load, the + items of a load in front of its - items, walked 64 at a time); a walk step is one ds_add_u32 of up to 64
lanes, serviced in two groups of 32 lanes; a group takes as many LDS cycles as its busiest bank has lanes (bank = dword address mod 32;
lanes on the SAME address serialise like lanes on the same bank -- MI355X_MICROARCH.md, LDS).  Reported: LDS cycles per diagram for the
votes alone, relative numbers only (the bins' record order on the device is the order of an atomic counter; row-major is assumed here).
    python tools/vote_bank_model.py [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import vote_cull_model as m  # noqa: E402
from img2sgf_amd import synth  # noqa: E402
from oracle import pipeline as opipe  # noqa: E402

EB, MINR, MAXR = 32, 1, 30


def lds_cycles(plane, VT, strides):
    """{stride: (LDS cycles, lane groups, votes)} for one plane; the items are formed once, the addresses per stride."""
    h, w = plane.shape
    x, y, sx, sy = m.records(plane)
    order = np.lexsort((x, y))                      # row-major inside a bin (assumption)
    x, y, sx, sy = x[order], y[order], sx[order], sy[order]
    binx, biny = x // EB, y // EB
    cycles = {v: 0 for v in strides}
    ideal = votes = 0
    steps = np.arange(MINR, MAXR + 1)[None, :]
    for ty in range(0, h, VT):
        for tx in range(0, w, VT):
            lx0, ly0 = tx - 1, ty - 1
            vx_lo, vy_lo = max(lx0, 0), max(ly0, 0)
            vx_hi, vy_hi = min(lx0 + VT + 2, w), min(ly0 + VT + 2, h)
            bx0, bx1 = max(lx0 - MAXR, 0) // EB, min(lx0 + VT + 1 + MAXR, w - 1) // EB
            by0, by1 = max(ly0 - MAXR, 0) // EB, min(ly0 + VT + 1 + MAXR, h - 1) // EB
            items = []
            for by in range(by0, by1 + 1):
                for bx in range(bx0, bx1 + 1):
                    sel = np.nonzero((binx == bx) & (biny == by))[0]
                    for k0 in range(0, len(sel), 64):
                        s = sel[k0:k0 + 64]
                        X0, Y0 = (x[s] - vx_lo) << 10, (y[s] - vy_lo) << 10
                        xl, yl = (vx_hi - vx_lo) << 10, (vy_hi - vy_lo) << 10
                        ax, bx_, ay, by_ = MINR * sx[s], MAXR * sx[s], MINR * sy[s], MAXR * sy[s]
                        mnx, mxx, mny, mxy = np.minimum(ax, bx_), np.maximum(ax, bx_), np.minimum(ay, by_), np.maximum(ay, by_)
                        in_p = (X0 + mxx >= 0) & (X0 + mnx < xl) & (Y0 + mxy >= 0) & (Y0 + mny < yl)
                        in_n = (X0 - mnx >= 0) & (X0 - mxx < xl) & (Y0 - mny >= 0) & (Y0 - mxy < yl)
                        items.append(np.stack([s[in_p], np.ones(in_p.sum(), np.int64)], 1))
                        items.append(np.stack([s[in_n], -np.ones(in_n.sum(), np.int64)], 1))
            if not items:
                continue
            it = np.concatenate(items)
            for k0 in range(0, len(it), 64):
                s, d = it[k0:k0 + 64, 0], it[k0:k0 + 64, 1]
                cx = ((x[s, None] << 10) + d[:, None] * steps * sx[s, None]) >> 10
                cy = ((y[s, None] << 10) + d[:, None] * steps * sy[s, None]) >> 10
                ok = (cx >= vx_lo) & (cx < vx_hi) & (cy >= vy_lo) & (cy < vy_hi)
                ry, rx = cy - ly0, cx - lx0
                for st in range(rx.shape[1]):
                    for g0 in (0, 32):
                        o = ok[g0:g0 + 32, st]
                        n = int(o.sum())
                        if not n:
                            continue
                        votes += n
                        ideal += 1
                        for v in strides:
                            cycles[v] += int(np.bincount((ry[g0:g0 + 32, st][o] * v + rx[g0:g0 + 32, st][o]) & 31, minlength=32).max())
    return cycles, ideal, votes


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    img = synth.synth_diagram(seed)[0]
    b = opipe.process_image(img)["blurs"]
    planes = [b[0], b[1], b[4], b[5], b[6], b[7], b[8], b[9]]
    print("seed %d, eight HoughCircles inputs; LDS cycles of the vote instructions (32-lane groups x busiest bank), model" % seed)
    for VT, strides in ((126, (129,)), (128, (131, 133, 135, 137, 139, 141, 143, 145))):
        tot, i, v = {k: 0 for k in strides}, 0, 0
        for pl in planes:
            c, gi, gv = lds_cycles(pl, VT, strides)
            i, v = i + gi, v + gv
            for k in strides:
                tot[k] += c[k]
        for k in strides:
            lds = (VT + 2) * k * 4
            print("  tile %d, row stride %d (%6d B of LDS for the tile): %9d votes, %8d lane groups, %8d LDS cycles = %.2f per group (1.00 = conflict-free)" % (
                VT, k, lds, v, i, tot[k], tot[k] / i))
