"""ONE GPU-less estimate of the vector instructions each hot kernel issues per diagram -- static instruction counts of
the compiled loops (hipcc -S, tools/isa_budget.py) x trip counts replayed on the oracle's planes of the benchmark's diagrams -- checked
against the counters of round 4 (profiles/r04_pmc_sq_clean_pass256.csv, SQ_INSTS_VALU; the kernels on the benchmark path have not
changed since), and then applied to every experiment under tools/experiments/: predicted change in vector instructions and, at the
kernel's MEASURED microseconds per instruction, in time.  The ranking decides the order of tools/experiments/ab.sh.

What is modelled, per kernel (w x h = 1024 x 1024; everything scales with the image):
  k_sobel_nms_rows<0|2>  wavefronts (7 planes x 4 column groups x 32 bands | 1 plane) x 36 input rows; per row the compiled row walks'
                         instruction count (byte walk on two-valued planes, packed walk on Gaussian planes) minus the blocks a row skips:
                         the border fix (waves without an image-edge lane), the suppression (rows with no magnitude above `low` in the
                         wavefront's 256 pixels: REPLAYED on the oracle's planes), the first rows' gradient / emit blocks, the worklist
  k_edge_bins            workgroups (128 x 32 pixels) x the fixed part + ceil(edge records of the block / 128) trips of the record loop per
                         wavefront (REPLAYED: the oracle's edge maps)
  k_vote_centres         record loads, surviving items and tiles from tools/vote_cull_model.py (which reproduces round 4's debug counters
                         to 0.3 %) x the compiled cull trip, walk (30 steps x 8) and zero / scan loops
  k_blur<true>           wavefronts x rows x the compiled row loop (two-valued speculation succeeds on every band of a clean diagram)
  k_radius, k_erase_lines  not modelled (latency-bound / data-dependent erase lists); k_radius' experiments are ranked by loads instead

    python tools/valu_model.py [-o profiles/r06_valu_model.md] [--seeds 0 1 2]
Uses the oracle (test infrastructure) for the planes; nothing in the product imports this."""
import argparse
import collections
import csv
import os
import re
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_budget as ib  # noqa: E402
from img2sgf_amd import build, synth  # noqa: E402

W = H = 1024
CR_R = 32


# ---------------------------------------------------------------------------------------------------------------------------------
# compiled code

def kernels_of(csrc):
    saved = build.CSRC, build.FLAGS
    build.FLAGS = [f if f != os.path.join(saved[0], "isa") else os.path.join(csrc, "isa") for f in build.FLAGS]
    build.CSRC = csrc
    try:
        with tempfile.TemporaryDirectory() as d:
            return ib.kernels(ib.compile_asm(d))
    finally:
        build.CSRC, build.FLAGS = saved


class Code:
    def __init__(self, body):
        self.ins, self.labels = [], {}
        for l in body.splitlines():
            l = l.strip()
            if not l or l.startswith(";"):
                continue
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                self.labels[m.group(1)] = len(self.ins)
            elif not l.startswith("."):
                self.ins.append(l.split(";")[0].strip())
        self.loops = ib.loops(body)[1]

    def valu(self, a, b):
        return sum(1 for x in self.ins[a:b] if x.startswith("v_"))

    def big_loops(self, least):
        """outermost loops with at least `least` vector instructions: [(first, last)] (loops that share a body count once)"""
        out = []
        for a, b in self.loops:
            if self.valu(a, b + 1) >= least and not any(a2 <= a and b <= b2 + 40 and (a2, b2) != (a, b) for a2, b2 in out):
                out.append((a, b))
        return out

    def blocks(self, a, b):
        """forward conditional skips inside [a, b]: (branch index, target index, vector instructions skipped, markers)"""
        out = []
        for i in range(a, b + 1):
            m = re.match(r"s_cbranch\w*\s+(\.LBB\w+)", self.ins[i])
            if m and i < self.labels.get(m.group(1), -1) <= b + 1:
                t = self.labels[m.group(1)]
                seg = self.ins[i + 1:t]
                out.append((i, t, sum(1 for x in seg if x.startswith("v_")),
                            {"atomic" if "atomic" in x else "perm" if x.startswith("v_perm") else "" for x in seg} - {""}))
        return out


def row_loop_facts(code, a, b):
    """One trip (6 rows) of a row walk: total vector instructions and the skippable blocks by kind.
    any  = the suppression, skipped when no pixel of the wavefront's row exceeds `low`: a skip of >= 20 vector instructions without
           atomics that sits inside a slightly larger skip -- emit = that wrapper (rows that produce no output: the first four of a band);
    grad = a skip of >= 40 without atomics and without such a wrapper / content (the gradient block: the first two rows of a band);
    fix  = skips of <= 6 (the border fix of wavefronts with an image-edge lane); worklist = outermost skips that hold an atomic."""
    T = code.valu(a, b + 1)
    blk = [x for x in code.blocks(a, b) if x[2] < 0.5 * T]           # (a skip of nearly the whole body is the loop's own exit test)
    plain = [x for x in blk if "atomic" not in x[3]]
    inside = lambda x, y: y[0] < x[0] and x[1] <= y[1] and x is not y
    f = dict(T=T, fix=0, grad=[], emit=[], any=[], worklist=0)
    used = set()
    for x in plain:
        if x[2] < 20:
            continue
        wrap = [y for y in blk if inside(x, y) and y[2] - x[2] <= 30]
        if wrap:
            w = min(wrap, key=lambda y: y[2])
            wl = sum(z[2] for z in blk if "atomic" in z[3] and inside(z, w) and not any(inside(z, q) and "atomic" in q[3] and inside(q, w) for q in blk))
            f["any"].append(x[2])
            f["emit"].append(w[2] - wl)
            used.add(x[0]); used.add(w[0])
    for x in plain:
        if x[0] in used or any(inside(x, y) for y in blk if y[0] in used):
            continue
        if x[2] <= 6:
            f["fix"] += x[2]
        elif x[2] >= 40 and not any(inside(y, x) and y[2] >= 20 for y in blk):
            f["grad"].append(x[2])
    for z in blk:
        if "atomic" in z[3] and z[0] not in used and not any(inside(z, q) and "atomic" in q[3] and q[0] not in used for q in blk):
            f["worklist"] += z[2]
    return f


# ---------------------------------------------------------------------------------------------------------------------------------
# replay on the oracle's planes

def sobel_mag(p):
    q = np.pad(p.astype(np.int32), 1, mode="edge")
    dx = (q[:-2, 2:] + 2 * q[1:-1, 2:] + q[2:, 2:]) - (q[:-2, :-2] + 2 * q[1:-1, :-2] + q[2:, :-2])
    dy = (q[2:, :-2] + 2 * q[2:, 1:-1] + q[2:, 2:]) - (q[:-2, :-2] + 2 * q[:-2, 1:-1] + q[:-2, 2:])
    return np.abs(dx) + np.abs(dy)


def replay(seeds):
    from oracle import cv_oracle as cvo, pipeline as opipe
    import vote_cull_model as vcm
    acc = collections.defaultdict(float)
    tile_items, above = {}, {}
    for s in seeds:
        img = synth.synth_diagram(s)[0]
        ref = opipe.process_image(img)
        b = ref["blurs"]
        planes = [b[0], b[1], b[4], b[5], b[6], b[7], b[8], b[9]]
        for v, p in enumerate(planes):
            m = (sobel_mag(p) > 50).reshape(H, W // 256, 256).any(axis=2)
            acc["f_any_v%d" % v] += m.mean()
            _, dbg = cvo.hough_circles(p, debug=True)
            e = dbg["edges"] != 0
            blocks = e.reshape(H // 32, 32, W // 128, 128).sum(axis=(1, 3))          # edge pixels per 128 x 32 block of k_edge_bins
            # record-loop trips: thread t takes records t, t + 128, ... : wavefront 0 makes ceil(n / 128) trips, wavefront 1 ceil((n - 64) / 128)
            acc["eb_trips"] += (np.ceil(blocks / 128.0) + np.ceil(np.maximum(blocks - 64, 0) / 128.0)).sum()
            acc["eb_records"] += e.sum()
            for VT in (126, 128):
                vcm.VT = VT
                r = vcm.model(p)
                acc["vote%d_loads" % VT] += r["today_batches"]
                acc["vote%d_items" % VT] += r["today_items"]
                acc["vote%d_notest" % VT] += r["today_batches_all_both"]
                acc["vote%d_bintrips" % VT] += r["bin_trips"]
                tile_items.setdefault((VT, v // 2), []).append(np.array(r["tile_items"]))
                # the final scan: a wavefront's trip covers two rows of the tile; it enters the candidate code if a cell of either input of the
                # pair exceeds the threshold there
                a_ = dbg["acc"][:H, :W] > 30
                above.setdefault((VT, v // 2), []).append(a_)
            vcm.VT = 126
        for (VT, pair), lst in tile_items.items():
            it = lst[0] + lst[1]                                       # items per workgroup (both inputs of the pair)
            rem = np.minimum(it, 16 * 31.5)                            # what the 16 wavefronts are left with (each < 64) and walk together at the end
            acc["vote%d_fullwalks" % VT] += ((it - rem) / 64.0).sum()
            acc["vote%d_remwalks" % VT] += np.ceil(rem / 64.0).sum()
            ab = above[(VT, pair)][0] | above[(VT, pair)][1]
            nt = -(-H // VT)
            hits = 0
            for ty in range(nt):
                for tx in range(nt):
                    t = ab[ty * VT:(ty + 1) * VT, tx * VT:(tx + 1) * VT]
                    rows = t.any(axis=1)
                    rows = np.pad(rows, (0, (-len(rows)) % 2))
                    hits += int(rows.reshape(-1, 2).any(axis=1).sum())
            acc["vote%d_scanhits" % VT] += hits
        tile_items.clear(); above.clear()
        acc["f_any_main"] += (sobel_mag(planes[0]) > 50).reshape(H, W // 256, 256).any(axis=2).mean()
    return {k: v / len(seeds) for k, v in acc.items()}


# ---------------------------------------------------------------------------------------------------------------------------------
# the models: vector wave-instructions per diagram

def sobel_model(code, rp, mode):
    big = code.big_loops(300)
    lb = [l for l in big if not any(x.startswith("v_pk_") for x in code.ins[l[0]:l[1] + 1])]
    lpk = [l for l in big if any(x.startswith("v_pk_") for x in code.ins[l[0]:l[1] + 1])]
    (a1, b1), (a2, b2) = lb[0], lpk[0]
    fb, fp = row_loop_facts(code, a1, b1), row_loop_facts(code, a2, b2)
    f_fix = 0.5                                                   # column groups 0 and 3 of 4 hold an image-edge lane
    waves = (W // 256) * (H // CR_R)                              # per plane
    sel = [l for l in code.loops if l[1] < a1 and 60 <= code.valu(l[0], l[1] + 1) <= 400]         # the border-fix selector loop (fix lanes only)
    v_sel = code.valu(sel[0][0], sel[0][1] + 1) if sel else 0
    # the byte walk's first trip is peeled in front of its loop: [peel, a1)
    peel = max([i for i in range(a1) if re.match(r"s_branch\s", code.ins[i]) and code.labels.get(code.ins[i].split()[1], 0) > b1] + [0])
    pre_common = code.valu(0, peel) - v_sel * (1 - f_fix)
    post_b = code.valu(b1 + 1, a2) * 0.5                          # epilogue of the byte walk: two-valued tests, the worklist (a few lanes)
    pre_p = code.valu(b1 + 1, a2) * 0.5

    def bytes_wave(f_any):
        trip = fb["T"] - (1 - f_fix) * fb["fix"] - (1 - f_any) * sum(fb["any"]) - fb["worklist"]
        peeled = code.valu(peel, a1) - (1 - f_fix) * fb["fix"] - (1 - f_any) * sum(x[2] for x in code.blocks(peel, a1 - 1) if 15 <= x[2] <= 40) * 0.5
        return pre_common + peeled + 5 * trip + post_b

    def packed_wave(f_any):
        # per trip of 6 rows: the worklist block runs once per band (1 row of 36), the two gradient blocks are skipped in the first trip only,
        # 4 of a band's 36 rows emit nothing, and the suppression runs on the emitting rows that hold a magnitude above `low`
        trip = (fp["T"] - (1 - f_fix) * fp["fix"] - fp["worklist"] * (35.0 / 36) - sum(fp["grad"]) / 6.0
                - (4.0 / 36) * 6 * np.mean(fp["emit"]) - (1 - f_any) * (32.0 / 36) * sum(fp["any"]))
        return pre_common + pre_p + 6 * trip

    out = {}
    if mode == 0:
        tot = 0.0
        for v in (1, 2, 4, 6):
            tot += waves * bytes_wave(rp["f_any_v%d" % v])
        for v in (3, 5, 7):
            tot += waves * packed_wave(rp["f_any_v%d" % v])
        out["valu"] = tot
        out["rows_bytes"], out["rows_packed"] = 4 * waves * 36, 3 * waves * 36
    else:
        out["valu"] = waves * bytes_wave(rp["f_any_main"])
        out["rows_bytes"], out["rows_packed"] = waves * 36, 0
    out["facts"] = dict(bytes_per_row=fb["T"] / 6.0, packed_per_row=fp["T"] / 6.0, bytes_any=np.mean(fb["any"]), packed_any=np.mean(fp["any"]))
    return out


def edge_bins_model(code, rp):
    # the record loop: the loop that holds the LDS append (ds_add_rtn) and the record store; out-of-line blocks behind s_endpgm make
    # spurious "loops" around the whole kernel, so it is found by content
    cand = [l for l in code.loops if any(x.startswith("ds_add_rtn") for x in code.ins[l[0]:l[1] + 1]) and not any(x.startswith("s_barrier") for x in code.ins[l[0]:l[1] + 1])]
    a, b = min(cand, key=lambda l: l[1] - l[0])
    blk = code.blocks(a, b)
    # the border pixels' byte path (sobel_at: byte loads) runs only when a lane's pixel touches the image border: the largest skip with global byte loads
    byte_path = max([x[2] for x in blk if x[2] < 0.5 * code.valu(a, b + 1) and any("load_ubyte" in y for y in code.ins[x[0]:x[1]])] + [0])
    body = code.valu(a, b + 1) - byte_path
    end = min([i for i, x in enumerate(code.ins) if x.startswith("s_endpgm")] + [len(code.ins)])
    # in front of the loop: one large skip, the mask of a block the image ENDS in (never at a width that is a multiple of 128)
    rare = sum(x[2] for x in code.blocks(0, a - 1) if x[2] >= 30 and x[1] <= a)
    fixed = code.valu(0, a) - rare + code.valu(b + 1, end)
    waves = (W // 128) * (H // 32) * 8 * 2
    return dict(valu=waves * fixed + rp["eb_trips"] * body, facts=dict(fixed_per_wave=fixed, record_trip=body, trips=rp["eb_trips"], byte_path=byte_path))


def vote_model(code, rp, VT=126):
    """k_vote_centres: per wavefront the prologue, the zeroing trips, the scan (4 trips of the workgroup over the tile's rows; the candidate
    code only where a cell exceeds the threshold: REPLAYED on the oracle's accumulators); per bin of a tile's window one trip of the bin
    loop; per started 64 records of a bin one trip of the cull; per 64 surviving items one walk of 30 steps; the wavefronts' remainders
    (< 64 items each) are walked together at the end (the second, per-step guarded copy of the walk)."""
    ins = code.ins
    bin_loop = min([l for l in code.loops if code.valu(l[0], l[1] + 1) >= 250 and any(x.startswith("ds_add_u32") for x in ins[l[0]:l[1] + 1])], key=lambda l: l[0])
    inner = [l for l in code.loops if bin_loop[0] < l[0] and l[1] <= bin_loop[1]]
    cull = max([l for l in inner if not any(x.startswith("ds_add_u32") for x in ins[l[0]:l[1] + 1]) and code.valu(l[0], l[1] + 1) >= 30], key=lambda l: l[1] - l[0])
    adds = [i for i, x in enumerate(ins) if x.startswith("ds_add_u32")]
    in_loop = [i for i in adds if bin_loop[0] <= i <= bin_loop[1]]
    after = [i for i in adds if i > bin_loop[1]]
    v_cull = code.valu(cull[0], cull[1] + 1)
    # an experiment may give the cull two paths (cull_fast: bins that need no reach test take a short trip that the compiler lays out as a
    # small loop of its own inside the cull loop)
    fast = test = v_cull
    sub = [l for l in code.loops if cull[0] < l[0] and l[1] < cull[1] and code.valu(l[0], l[1] + 1) < 0.4 * v_cull]
    if sub:
        n_fast = max(code.valu(l[0], l[1] + 1) for l in sub)
        fast, test = n_fast + 5, v_cull - n_fast
    v_walk = code.valu(cull[1] + 1, in_loop[-1] + 1) + 6
    v_bin = code.valu(bin_loop[0], bin_loop[1] + 1) - v_cull - v_walk + 6
    # the remainder walk: between the two workgroup barriers behind the bin loop, from its first lane guard on
    bars = [i for i, x in enumerate(ins) if x.startswith("s_barrier") and i > bin_loop[1]]
    guard = min(i for i in range(bars[0], bars[1]) if ins[i].startswith("s_cbranch_execz") and i > bars[0] + 10)
    v_rem = code.valu(guard, bars[1])
    scan = max([l for l in code.loops if l[0] > after[-1]], key=lambda l: l[1] - l[0])
    scan_blk = [x for x in code.blocks(scan[0], scan[1]) if x[2] >= 0.6 * code.valu(scan[0], scan[1] + 1)]
    v_scan_hit = max([x[2] for x in scan_blk] + [0])
    v_scan_base = code.valu(scan[0], scan[1] + 1) - v_scan_hit
    v_fixed = code.valu(0, bin_loop[0]) + 3 * 10 + (code.valu(bin_loop[1] + 1, scan[0]) - v_rem) + v_scan_base * -(-VT * 32 // 1024)
    tiles = (-(-W // VT)) * (-(-H // VT)) * 4                      # workgroups per diagram (pairs of inputs)
    loads, notest = rp["vote%d_loads" % VT], rp["vote%d_notest" % VT]
    valu = (tiles * 16 * v_fixed + rp["vote%d_bintrips" % VT] * v_bin + (loads - notest) * test + notest * fast
            + rp["vote%d_fullwalks" % VT] * v_walk + rp["vote%d_remwalks" % VT] * v_rem + rp["vote%d_scanhits" % VT] * v_scan_hit)
    return dict(valu=valu, facts=dict(fixed_per_wave=v_fixed, bin_trip=v_bin, cull_trip=(fast, test), walk=v_walk, remainder_walk=v_rem, scan_hit=v_scan_hit,
                                      loads=loads, no_test_loads=notest, bin_trips=rp["vote%d_bintrips" % VT], full_walks=rp["vote%d_fullwalks" % VT],
                                      remainder_walks=rp["vote%d_remwalks" % VT], scan_hits=rp["vote%d_scanhits" % VT], workgroups=tiles))


def blur_model(code):
    """k_blur<true> on clean diagrams: every band passes the two-valued speculation.  The row loop is unrolled by the ring depth (7 rows per
    trip); per row one skip of <= 30 vector instructions (the border fix: wavefronts with an image-edge lane, half of them at 1024) and one
    of 40 - 60 (forming the output bytes: not on the 6 apron rows of the 70 a band reads); in front of the loop one large skip -- the border
    selectors, fix wavefronts only."""
    f_fix = 0.5
    a, b = max(code.big_loops(600), key=lambda l: l[0])              # the innermost of the loops that share the body
    T = code.valu(a, b + 1)
    blk = [x for x in code.blocks(a, b) if x[2] < 0.5 * T]
    top = [x for x in blk if not any(y[0] < x[0] and x[1] <= y[1] and y is not x for y in blk)]
    rows_per_trip = sum(1 for x in top if 40 <= x[2] <= 60) or 7
    fix = sum(x[2] for x in top if x[2] <= 30 and "perm" in x[3])
    outp = sum(x[2] for x in top if 40 <= x[2] <= 60)
    per_row = (T - (1 - f_fix) * fix - (6.0 / 70) * outp) / rows_per_trip
    pre_blocks = [x for x in code.blocks(0, a - 1) if x[2] >= 100 and x[1] <= a]
    pre = code.valu(0, a) - (1 - f_fix) * sum(x[2] for x in pre_blocks if not any(y[0] < x[0] and x[1] <= y[1] and y is not x for y in pre_blocks))
    waves = (W // 256) * (H // 64)
    return dict(valu=waves * (pre + 70 * per_row), facts=dict(per_row=per_row, rows_per_trip=rows_per_trip, prologue=pre))


# ---------------------------------------------------------------------------------------------------------------------------------

def measured():
    out = {}
    with open(os.path.join(ROOT, "profiles", "r04_pmc_sq_clean_pass256.csv")) as f:
        for r in csv.DictReader(f):
            out[r["kernel"]] = float(r["SQ_INSTS_VALU_per_dispatch"]) * float(r["dispatches"]) / 256.0
    us = {}
    with open(os.path.join(ROOT, "profiles", "r04_clean_kernel_stats.csv")) as f:
        for r in csv.DictReader(f):
            us[r["kernel"]] = float(r["total_us"]) / float(r["calls"]) / 256.0
    return out, us


def model_all(ks, rp, VT=126):
    res = {}
    res["k_sobel_nms_rows<0, true>"] = sobel_model(Code(ks["k_sobel_nms_rows<0, true>"]["body"]), rp, 0)
    res["k_sobel_nms_rows<2, true>"] = sobel_model(Code(ks["k_sobel_nms_rows<2, true>"]["body"]), rp, 2)
    res["k_edge_bins"] = edge_bins_model(Code(ks["k_edge_bins"]["body"]), rp)
    res["k_vote_centres<30>"] = vote_model(Code(ks["k_vote_centres<30>"]["body"]), rp, VT)
    res["k_blur<true>"] = blur_model(Code(ks["k_blur<true>"]["body"]))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-o")
    ap.add_argument("--seeds", type=int, nargs="*", default=[0, 1, 2])
    ap.add_argument("--experiments", nargs="*", default=["stride133", "tile128", "tile128+cull_fast", "tile128+vastr133", "canny_lean", "edge_lut", "radius_pre2", "radius_staged"])
    a = ap.parse_args()
    rp = replay(a.seeds)
    meas, us = measured()
    base = model_all(kernels_of(build.CSRC), rp)
    L = []
    L.append("| kernel | model: vector wave-instructions per diagram | round 4's counter | model / counter | µs per diagram (round 4) | µs per 1 000 instructions |")
    L.append("|---|---|---|---|---|---|")
    for k, r in base.items():
        L.append("| `%s` | %.0f | %.0f | %.3f | %.2f | %.4f |" % (k, r["valu"], meas[k], r["valu"] / meas[k], us[k], us[k] / meas[k] * 1000))
    L.append("")
    L.append("| experiment | kernel | Δ vector instructions per diagram (model) | of the kernel | predicted Δ µs per diagram at the kernel's measured µs per instruction |")
    L.append("|---|---|---|---|---|")
    rank = []
    for name in a.experiments:
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "experiments", "apply.py"), name], stdout=subprocess.DEVNULL)
        ks = kernels_of(os.path.join(ROOT, "build", "exp", name, "pkg", "csrc"))
        new = model_all(ks, rp, VT=128 if "tile128" in name else 126)
        tot = 0.0
        for k in base:
            d = new[k]["valu"] - base[k]["valu"]
            if abs(d) > 0.002 * base[k]["valu"]:
                dus = d * us[k] / meas[k]
                tot += dus
                L.append("| `%s` | `%s` | %+.0f | %+.1f %% | %+.2f |" % (name, k, d, 100.0 * d / base[k]["valu"], dus))
        rank.append((tot, name))
    L.append("")
    L.append("Ranking by predicted Δ µs per diagram (vector-instruction model only; `stride133` / `vastr133` change LDS bank cycles, `radius_*` change loads -- see their own models):")
    for tot, name in sorted(rank):
        L.append("* `%s`: %+.2f µs" % (name, tot))
    text = "\n".join(L)
    print(text)
    print("\nreplay:", {k: round(v, 3) for k, v in rp.items()})
    for k, r in base.items():
        print(k, {kk: (round(float(vv), 1) if isinstance(vv, (int, float, np.floating, np.integer)) else vv) for kk, vv in r["facts"].items()})
    if a.o:
        with open(a.o, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
