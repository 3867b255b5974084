#!/usr/bin/env python3
"""Every kernel dispatch of a rocprofv3 rocpd database in launch order: index, kernel, grid size, start (us since the first dispatch),
duration (us).  Usage: rocpd_sequence.py DB OUT.csv   -- the per-dispatch companion of rocpd_stats.py: tools/roofline_md.py needs the order
to tell the main Canny's hysteresis launches from HoughCircles' (same kernels, different phases)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"^void ", "", name).replace("i2s::", "")


def main():
    cur = sqlite3.connect(sys.argv[1]).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    grid = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
    rows = cur.execute("select %s, start, end%s from kernels order by start" % (name_col, ", " + grid if grid else "")).fetchall()
    t0 = rows[0][1] if rows else 0
    with open(sys.argv[2], "w") as f:
        f.write("index,kernel,grid_x,start_us,duration_us\n")
        for i, r in enumerate(rows):
            n = short(r[0])
            n = '"%s"' % n if "," in n else n
            f.write("%d,%s,%s,%.2f,%.2f\n" % (i, n, r[3] if grid else "", (r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3))
    print("wrote", sys.argv[2], len(rows), "dispatches")


if __name__ == "__main__":
    main()
