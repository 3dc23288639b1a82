"""Launch-ordered kernel trace of the LAST 1/k of a rocprofv3 --kernel-trace database (k = number of identical iterations the traced command ran), one row per
launch: the input of per-layer tables (tools/vae_layer_table.py).
Usage: python tools/export_trace.py <results.db> <out.csv> [k]"""
import csv
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
k = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cur = sqlite3.connect(db).cursor()
kcols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
gcol = next((c for c in kcols if c.lower() in ("grid_size_x", "grid_size", "grid_x")), None)
rows = cur.execute(f"select name, start, end, {gcol or 0} from kernels order by start").fetchall()
n = len(rows) // k
rows = rows[len(rows) - n:]
t0 = rows[0][1]
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["index", "kernel", "grid_threads_x", "start_us", "duration_us"])
    for i, (name, s, e, g) in enumerate(rows):
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        w.writerow([i, name, g, round((s - t0) / 1e3, 1), round((e - s) / 1e3, 1)])
print(f"{out}: {len(rows)} launches, {(rows[-1][2] - t0) / 1e6:.2f} ms from the first start to the last end")
