"""A CPU model of k_radius' record loads (counted before building).  For the synthetic workload's diagrams it rebuilds,
per HoughCircles input, the edge records binned 32 x 32 as k_edge_bins lays them out and the centre candidates k_vote_centres emits
(accumulator local maxima above the threshold, from the oracle's accumulator), and counts the record bytes k_radius pulls:

  today    one wavefront per centre: for each of the <= 3 x 3 bins within max_r of the centre, 64 slots unconditionally (the prefetch
           that does not wait for the counts) and the rest of the bin in batches of 64; 8-byte records, of which .x is used (the
           loads touch every 64-byte line of the records all the same)
  staged   centres grouped by their OWN 32 x 32 bin (all centres of a bin share the same 3 x 3 neighbourhood, up to the image border):
           one workgroup takes a bin's run of centres and brings the neighbourhood's records into LDS once

    python tools/radius_load_model.py [seeds...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from img2sgf_amd import synth  # noqa: E402
from oracle import cv_oracle as cvo, pipeline as opipe  # noqa: E402

EB, MINR, MAXR, THR, CHUNK = 32, 1, 30, 30, 8


def centres(acc, w, h):
    a = acc.astype(np.int64)
    c = a[1:h, 1:w]                                    # cells (x, y), 1 <= x <= w-1, 1 <= y <= h-1
    m = (c > THR) & (c > a[1:h, 0:w - 1]) & (c >= a[1:h, 2:w + 1]) & (c > a[0:h - 1, 1:w]) & (c >= a[2:h + 1, 1:w])
    ys, xs = np.nonzero(m)
    return xs + 1, ys + 1


def model(plane, noisy=False):
    h, w = plane.shape
    _, dbg = cvo.hough_circles(plane, debug=True)
    ey, ex = np.nonzero(dbg["edges"])
    # records exist for edge pixels with a non-zero gradient (k_edge_bins drops the others): as in tools/vote_cull_model.py
    p = np.pad(plane.astype(np.int32), 1, mode="edge")
    dx = (p[ey, ex + 2] + 2 * p[ey + 1, ex + 2] + p[ey + 2, ex + 2]) - (p[ey, ex] + 2 * p[ey + 1, ex] + p[ey + 2, ex])
    dy = (p[ey + 2, ex] + 2 * p[ey + 2, ex + 1] + p[ey + 2, ex + 2]) - (p[ey, ex] + 2 * p[ey, ex + 1] + p[ey, ex + 2])
    keep = (dx != 0) | (dy != 0)
    ex, ey = ex[keep], ey[keep]
    bw, bh = -(-w // EB), -(-h // EB)
    cnt = np.zeros((bh, bw), np.int64)
    np.add.at(cnt, (ey // EB, ex // EB), 1)
    cx, cy = centres(dbg["acc"], w, h)
    assert len(cx) == dbg["n_centers"], (len(cx), dbg["n_centers"])
    out = dict(centres=len(cx), records=len(ex), today_slots=0, today_records_used=0, staged_records=0, groups=0, lds_records_max=0,
               today_bin_visits=0, staged_bin_visits=0)
    bx0 = np.maximum(cx - MAXR, 0) // EB
    bx1 = np.minimum(cx + MAXR + 1, w - 1) // EB
    by0 = np.maximum(cy - MAXR, 0) // EB
    by1 = np.minimum(cy + MAXR + 1, h - 1) // EB
    for i in range(len(cx)):
        win = cnt[by0[i]:by1[i] + 1, bx0[i]:bx1[i] + 1]
        out["today_slots"] += int((np.maximum(64, -(-win // 64) * 64)).sum())
        out["today_records_used"] += int(win.sum())
        out["today_bin_visits"] += win.size
    # grouped by the centre's own bin; the group's window = the union of its centres' windows
    key = (cy // EB) * bw + (cx // EB)
    for k in np.unique(key):
        m = key == k
        x0, x1, y0, y1 = bx0[m].min(), bx1[m].max(), by0[m].min(), by1[m].max()
        win = cnt[y0:y1 + 1, x0:x1 + 1]
        out["staged_records"] += int(win.sum())
        out["staged_bin_visits"] += win.size
        out["groups"] += 1
        out["lds_records_max"] = max(out["lds_records_max"], int(win.sum()))
    # the experiment as built (tools/experiments/radius_staged.patch): the list sorted by (bin, position), a wavefront takes 8 consecutive
    # entries and stages a neighbourhood whenever the window of bins differs from the one it holds (a new chunk starts with nothing staged
    # only for the wavefront's first chunk, but chunks 128 apart never share a window in practice: counted as a reload)
    order = np.lexsort((cx, cy, key))
    wk = (bx0 | by0 << 10 | (bx1 - bx0) << 20 | (by1 - by0) << 22)[order]
    out["patch_records"] = 0
    out["patch_stagings"] = 0
    for c0 in range(0, len(order), CHUNK):
        prev = -1
        for j in range(c0, min(c0 + CHUNK, len(order))):
            if wk[j] != prev:
                i = order[j]
                out["patch_records"] += int(cnt[by0[i]:by1[i] + 1, bx0[i]:bx1[i] + 1].sum())
                out["patch_stagings"] += 1
                prev = wk[j]
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--noisy"]
    noisy = "--noisy" in sys.argv
    seeds = [int(a) for a in args] or [0]
    tot = None
    for s in seeds:
        img = synth.synth_diagram(s, noisy=noisy)[0]
        ref = opipe.process_image(img)
        b = ref["blurs"]
        for v, plane in enumerate([b[0], b[1], b[4], b[5], b[6], b[7], b[8], b[9]]):
            r = model(plane)
            print("seed %d input %d: %s" % (s, v, r))
            tot = r if tot is None else {k: (max(tot[k], r[k]) if k.endswith("_max") else tot[k] + r[k]) for k in r}
    n = len(seeds)
    print("\nper diagram (8 inputs):")
    for k, v in tot.items():
        print("  %-22s %12.0f" % (k, v if k.endswith("_max") else v / n))
    t, s_ = tot["today_slots"] / n * 8, tot["staged_records"] / n * 8
    print("  today : %.2f MiB of record slots per diagram (%.1f centres per diagram, %.0f slots per centre, %.0f%% of the slots hold a record)"
          % (t / 2**20, tot["centres"] / n, tot["today_slots"] / tot["centres"], 100.0 * tot["today_records_used"] / tot["today_slots"]))
    print("  staged: %.2f MiB per diagram (%.0f groups, %.1f centres per group, largest neighbourhood %d records = %d KB of LDS at 4 B per record)"
          % (s_ / 2**20, tot["groups"] / n, tot["centres"] / tot["groups"], tot["lds_records_max"], tot["lds_records_max"] * 4 // 1024 + 1))
    p_ = tot["patch_records"] / n * 8
    print("  patch : %.2f MiB per diagram (%.0f stagings: sorted list, chunks of %d entries per wavefront, restaged when the window of bins changes)"
          % (p_ / 2**20, tot["patch_stagings"] / n, CHUNK))
    print("  ratio : ideal grouping %.2f x, the patch %.2f x fewer record bytes than today" % (t / s_, t / p_))
