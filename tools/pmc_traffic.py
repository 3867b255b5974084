#!/usr/bin/env python3
"""HBM traffic of the blur+Canny stage from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE need separate passes: the
TCC block has 4 counter slots, FETCH_SIZE takes 3, WRITE_SIZE 2) of ONE device pass of the benchmark workload:

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o r -- python tools/kernel_times.py --images 128 --reps 1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o r -- python tools/kernel_times.py --images 128 --reps 1
  python tools/pmc_traffic.py gpurun_out/pmc_fetch/r_results.db gpurun_out/pmc_write/r_results.db 128 profiles/traffic.json

Units and corrections as MI355X_MICROARCH.md (HBM section) prescribes: both counters report KiB; on gfx950 FETCH_SIZE reports
exactly half of the bytes of wide coalesced streaming reads (x2), WRITE_SIZE is taken as is.  The stage's dispatches: k_grey,
k_blur (or k_median3 + k_gauss357), k_median57, the main-Canny Sobel/NMS dispatch (the smaller of the two
k_sobel_nms_rows<1 | 2 | 3> grids) and the hysteresis launches that precede the HoughCircles Sobel/NMS dispatch.
The output records the hash of the kernel sources (bench.py refuses the figure when the kernels have changed since)."""
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGE = ("k_grey", "k_blur", "k_median3", "k_gauss357", "k_median57_bin", "k_median57")


# the sources of the blur+Canny stage's kernels (and what they include)
STAGE_SOURCES = ("i2s_types.h", "k_canny.h", "k_canny_rows.h", "k_filters.h", "tile_io.h", "isa/gfx950_ops.h")


def kernels_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "img2sgf_amd", "csrc")
    for f in STAGE_SOURCES:
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "", name)
    return re.sub(r"^void ", "", name).replace("i2s::", "")


def stage_counts(db_path, counter):
    """{label: KiB} summed over the stage's dispatches of one rocpd database."""
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    kname = "kernel_name" if "kernel_name" in ix else "name"
    rows = [(r[ix["dispatch_id"]], short(r[ix[kname]]), int(r[ix["grid_size_x"]]), r[ix["counter_name"]], float(r[ix["value"]]))
            for r in cur.execute("select * from counters_collection")]
    rows = [r for r in rows if r[3] == counter]
    disp = {}
    for d, k, gsz, _, v in rows:
        e = disp.setdefault(d, [k, gsz, 0.0])
        e[2] += v
    order = sorted(disp)
    sobel = [d for d in order if disp[d][0] == "k_sobel_nms_rows"]
    hc_sobel = max(sobel, key=lambda d: disp[d][1]) if len(sobel) > 1 else None       # the 7-plane HoughCircles dispatch
    out = {}
    for d in order:
        k, gsz, v = disp[d]
        if k in STAGE:
            out[k] = out.get(k, 0.0) + v
        elif k == "k_sobel_nms_src" or (k == "k_sobel_nms_rows" and d != hc_sobel):
            out["k_sobel_nms(main Canny)"] = out.get("k_sobel_nms(main Canny)", 0.0) + v
        elif k == "k_hysteresis" and (hc_sobel is None or d < hc_sobel):
            out["k_hysteresis(main Canny)"] = out.get("k_hysteresis(main Canny)", 0.0) + v
    return out


def canny7_counts(db_path, counter):
    """KiB of the HoughCircles x7 Sobel/NMS dispatch and the hysteresis launches behind it (up to k_edge_bins)."""
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    ix = {c: i for i, c in enumerate(cols)}
    kname = "kernel_name" if "kernel_name" in ix else "name"
    disp = {}
    for r in cur.execute("select * from counters_collection"):
        if r[ix["counter_name"]] != counter:
            continue
        e = disp.setdefault(r[ix["dispatch_id"]], [short(r[ix[kname]]), int(r[ix["grid_size_x"]]), 0.0])
        e[2] += float(r[ix["value"]])
    order = sorted(disp)
    sobel = [d for d in order if disp[d][0] == "k_sobel_nms_rows"]
    if len(sobel) < 2:
        return None
    hc = max(sobel, key=lambda d: disp[d][1])
    bins = [d for d in order if disp[d][0] == "k_edge_bins" and d > hc]
    end = bins[0] if bins else order[-1] + 1
    return sum(disp[d][2] for d in order if d == hc or (hc < d < end and disp[d][0] == "k_hysteresis"))


def main():
    fetch_db, write_db, images, out_path = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fetch = stage_counts(fetch_db, "FETCH_SIZE")
    write = stage_counts(write_db, "WRITE_SIZE")
    total = sum(2.0 * v for v in fetch.values()) * 1024 + sum(write.values()) * 1024
    n_pix = 1024 * 1024
    f7, w7 = canny7_counts(fetch_db, "FETCH_SIZE"), canny7_counts(write_db, "WRITE_SIZE")
    doc = {
        "source": "two rocprofv3 --kernel-trace --pmc passes (FETCH_SIZE; WRITE_SIZE) of `python tools/kernel_times.py --images %d "
                  "--reps 1` (one device pass of the benchmark workload), reduced by tools/pmc_traffic.py" % images,
        "unit_note": "FETCH_SIZE / WRITE_SIZE in KiB; gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x 2, WRITE_SIZE as is",
        "kernels_sha": kernels_sha(),
        "images": images,
        "blur_canny_stage_kib_per_pass": {"fetch_raw": fetch, "write": write},
        "blur_canny_hbm_bytes_per_image": total / images,
        "canny7_kib_per_pass": {"fetch_raw": f7, "write": w7},
        "canny7_hbm_bytes_per_image": None if f7 is None else (2.0 * f7 + w7) * 1024 / images,
        "canny7_algorithmic_bytes_per_image": 14 * n_pix,
        "algorithmic_bytes_per_image_unfused": 14 * n_pix,
        "ratio_to_algorithmic": total / images / (14 * n_pix),
    }
    with open(out_path, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
