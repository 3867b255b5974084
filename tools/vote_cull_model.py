"""A CPU model of k_vote_centres' cull (counted before building).  For the synthetic workload's diagrams it rebuilds
the edge records (x, y, sx, sy) of every HoughCircles input from the oracle's edge maps, lays them out in 32 x 32 bins as k_edge_bins
does, and replays what each 126 x 126 accumulator tile's workgroup does with its 7 x 7 window of bins:

  today      every bin of the window is loaded in 64-record batches; every lane runs the two-direction reach test
  octants    each bin keeps 8 sub-lists by gradient octant (sign sx, sign sy, |sx| > |sy|); per (bin, octant, direction) a group-level
             classification from the bin's rectangle and the octant's cone: cannot reach -> skipped before the load; surely reaches ->
             taken without the per-lane test; straddling -> today's per-lane test (one direction)

and counts batches, lane utilisation and an instruction estimate for both.  Uses the oracle (test infrastructure) for the edge maps:
    python tools/vote_cull_model.py [seeds...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img2sgf_amd import synth  # noqa: E402
from oracle import cv_oracle as cvo, pipeline as opipe  # noqa: E402

EB, VT, MINR, MAXR = 32, 126, 1, 30


def records(plane):
    _, dbg = cvo.hough_circles(plane, debug=True)
    ys, xs = np.nonzero(dbg["edges"])
    p = np.pad(plane.astype(np.int32), 1, mode="edge")
    dx = (p[ys, xs + 2] + 2 * p[ys + 1, xs + 2] + p[ys + 2, xs + 2]) - (p[ys, xs] + 2 * p[ys + 1, xs] + p[ys + 2, xs])
    dy = (p[ys + 2, xs] + 2 * p[ys + 2, xs + 1] + p[ys + 2, xs + 2]) - (p[ys, xs] + 2 * p[ys, xs + 1] + p[ys, xs + 2])
    keep = (dx != 0) | (dy != 0)
    xs, ys, dx, dy = xs[keep], ys[keep], dx[keep].astype(np.float32), dy[keep].astype(np.float32)
    mag = np.sqrt(dx * dx + dy * dy)
    sx = np.rint(dx * np.float32(1024) / mag).astype(np.int64)
    sy = np.rint(dy * np.float32(1024) / mag).astype(np.int64)
    return xs.astype(np.int64), ys.astype(np.int64), sx, sy


def model(plane):
    h, w = plane.shape
    x, y, sx, sy = records(plane)
    octant = (sx < 0) * 4 + (sy < 0) * 2 + (np.abs(sx) > np.abs(sy)) * 1
    binx, biny = x // EB, y // EB
    out = dict(records=len(x), today_batches=0, today_lanes=0, today_items=0, oct_batches_test=0, oct_batches_all=0, oct_lanes=0,
               oct_groups=0, oct_groups_skipped=0, oct_items=0, today_batches_all_both=0, bin_trips=0, tile_items=[])
    for ty in range(0, h, VT):
        for tx in range(0, w, VT):
            lx0, ly0 = tx - 1, ty - 1
            vx_lo, vy_lo = max(lx0, 0), max(ly0, 0)
            vx_hi, vy_hi = min(lx0 + VT + 2, w), min(ly0 + VT + 2, h)        # valid cells [lo, hi)
            bx0, bx1 = max(lx0 - MAXR, 0) // EB, min(lx0 + VT + 1 + MAXR, w - 1) // EB
            by0, by1 = max(ly0 - MAXR, 0) // EB, min(ly0 + VT + 1 + MAXR, h - 1) // EB
            out["bin_trips"] += (bx1 - bx0 + 1) * (by1 - by0 + 1)          # the workgroup's bin loop visits every bin of the window, empty or not
            sel = (binx >= bx0) & (binx <= bx1) & (biny >= by0) & (biny <= by1)
            X, Y, SX, SY, O, BX, BY = x[sel], y[sel], sx[sel], sy[sel], octant[sel], binx[sel], biny[sel]
            # the kernel's per-lane test, both directions (fixed point, as in k_hough_circles.h)
            X0, Y0 = (X - vx_lo) << 10, (Y - vy_lo) << 10
            xl, yl = (vx_hi - vx_lo) << 10, (vy_hi - vy_lo) << 10
            ax, bx_, ay, by_ = MINR * SX, MAXR * SX, MINR * SY, MAXR * SY
            mnx, mxx, mny, mxy = np.minimum(ax, bx_), np.maximum(ax, bx_), np.minimum(ay, by_), np.maximum(ay, by_)
            in_p = (X0 + mxx >= 0) & (X0 + mnx < xl) & (Y0 + mxy >= 0) & (Y0 + mny < yl)
            in_n = (X0 - mnx >= 0) & (X0 - mxx < xl) & (Y0 - mny >= 0) & (Y0 - mxy < yl)
            out["tile_items"].append(int(in_p.sum() + in_n.sum()))         # (tools/valu_model.py: walks per workgroup)
            key = (BY - by0) * 8 + (BX - bx0)
            for k in np.unique(key):
                m = key == k
                n = int(m.sum())
                if bool(in_p[m].all() and in_n[m].all()):
                    out["today_batches_all_both"] = out.get("today_batches_all_both", 0) + -(-n // 64)
                out["today_batches"] += -(-n // 64)
                out["today_lanes"] += n
                out["today_items"] += int(in_p[m].sum() + in_n[m].sum())
                for o in range(8):
                    mo = m & (O == o)
                    no = int(mo.sum())
                    if not no:
                        continue
                    for d, inn in ((0, in_p), (1, in_n)):
                        out["oct_groups"] += 1
                        got = int(inn[mo].sum())
                        # the group-level classification is at best as sharp as "no record reaches" / "every record reaches"
                        if got == 0:
                            out["oct_groups_skipped"] += 1
                        else:
                            out["oct_items"] += got
                            if got == no:
                                out["oct_batches_all"] += -(-no // 64)
                            else:
                                out["oct_batches_test"] += -(-no // 64)
                            out["oct_lanes"] += no
    return out


if __name__ == "__main__":
    seeds = [int(a) for a in sys.argv[1:]] or [0]
    tot = None
    for s in seeds:
        img = synth.synth_diagram(s)[0]
        ref = opipe.process_image(img)
        b = ref["blurs"]
        for v, plane in enumerate([b[0], b[1], b[4], b[5], b[6], b[7], b[8], b[9]]):
            r = model(plane)
            print("seed %d input %d: %s" % (s, v, r))
            r.pop("tile_items")
            tot = r if tot is None else {k: tot[k] + r[k] for k in r}
    n = len(seeds)
    print("\nper diagram (8 inputs):")
    for k, v in tot.items():
        print("  %-20s %10.0f" % (k, v / n))
    tb, tl = tot["today_batches"] / n, tot["today_lanes"] / n
    print("  today: %.0f batches, %.1f records per batch (of 64), %.2f items per loaded record" % (tb, tl / tb, tot["today_items"] / tot["today_lanes"]))
    print("  today: %.0f batches (%.0f%%) lie in bins all of whose records reach in BOTH directions (no test needed)" % (
        tot["today_batches_all_both"] / n, 100.0 * tot["today_batches_all_both"] / tot["today_batches"]))
    ob = (tot["oct_batches_test"] + tot["oct_batches_all"]) / n
    print("  octant sub-lists, one batch per (bin, octant, direction) group at best-case classification: %.0f batches (%.0f test + %.0f all), "
          "%.1f records per batch, %.0f%% of the groups skipped" % (ob, tot["oct_batches_test"] / n, tot["oct_batches_all"] / n,
                                                                      tot["oct_lanes"] / n / ob, 100.0 * tot["oct_groups_skipped"] / tot["oct_groups"]))
