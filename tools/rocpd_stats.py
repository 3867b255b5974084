#!/usr/bin/env python3
"""Per-kernel summary (calls, total / average / min / max duration) from a rocprofv3 rocpd SQLite database
(`rocprofv3 --kernel-trace ... -d DIR -o NAME` writes DIR/NAME_results.db).  Usage: rocpd_stats.py DB [OUT.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"^void ", "", name)
    return name.replace("i2s::", "")


def q(name):
    """CSV field: template arguments carry commas (k_circles_final<4096, 2048>)."""
    return '"%s"' % name.replace('"', '""') if ("," in name or '"' in name) else name


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    lines = ["kernel,calls,total_us,avg_us,min_us,max_us,percent"]
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append("%s,%d,%.1f,%.2f,%.2f,%.2f,%.2f" % (q(n), a[0], a[1] / 1e3, a[1] / a[0] / 1e3, a[2] / 1e3, a[3] / 1e3,
                                                         100.0 * a[1] / total))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
