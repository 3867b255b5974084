#!/usr/bin/env python3
"""Per-kernel PMC counter sums from a rocprofv3 rocpd database (run with --kernel-trace --pmc ...).
Usage: rocpd_pmc.py DB [OUT.csv]   -> kernel, calls, <counter> per dispatch (mean) ..."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    return re.sub(r"^void ", "", name).replace("i2s::", "")


def q(name):
    """CSV field: template arguments carry commas (k_circles_final<4096, 2048>)."""
    return '"%s"' % name.replace('"', '""') if ("," in name or '"' in name) else name


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    rows = cur.execute("select * from counters_collection").fetchall()
    ix = {c: i for i, c in enumerate(cols)}
    kname = "kernel_name" if "kernel_name" in ix else "name"
    cname = "counter_name" if "counter_name" in ix else "name"
    agg, names = {}, []
    disp = {}
    for r in rows:
        k, c, v = short(r[ix[kname]]), r[ix[cname]], float(r[ix["value"]])
        if c not in names:
            names.append(c)
        agg.setdefault(k, {}).setdefault(c, 0.0)
        agg[k][c] += v
        disp.setdefault(k, set()).add(r[ix["dispatch_id"]])
    lines = ["kernel,dispatches," + ",".join(n + "_per_dispatch" for n in names)]
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        n = len(disp[k])
        lines.append(q(k) + "," + str(n) + "," + ",".join("%.6g" % (agg[k].get(c, 0.0) / n) for c in names))
    txt = "\n".join(lines)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
