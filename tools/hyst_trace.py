import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = cur.execute("select %s, start, end from kernels order by start" % name_col).fetchall()
out = []
for n, s, e in rows:
    n = re.sub(r"\(.*", "", n).replace("void ", "").replace("i2s::", "")
    if "hyst" in n or "sobel" in n or "edge_bins" in n: out.append("%s %.1f" % (n.split("<")[0][:18], (e - s) / 1e3))
print(" | ".join(out[-40:]))
