"""Turn a rocprofv3 (--kernel-trace [--stats]) sqlite database into the per-kernel summary committed under profiles/.
Usage: python tools/export_profile.py <results.db> <out.csv> [steps]   (durations are divided by `steps` when given)"""
import csv
import re
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
cur = sqlite3.connect(db).cursor()
kcols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
gcol = next((c for c in kcols if c.lower() in ("grid_size_x", "grid_size", "grid_x")), None)
# one row per (kernel, launch grid): the self-attention launches (grid 8192 x 256 threads) and the cross-attention launches
# (grid 768) of the same kernel are different workloads and get their own lines
gsel, ggrp = (f"{gcol}", f", {gcol}") if gcol else ("0", "")
rows = cur.execute(f"select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), max(vgpr_count), max(accum_vgpr_count), "
                   f"max(lds_size), {gsel} from kernels group by name{ggrp} order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid_threads_x", "calls_per_step", "total_ms_per_step", "percent", "avg_us", "min_us", "max_us", "arch_vgpr", "accum_vgpr", "lds_bytes"])
    for n, c, s, a, mn, mx, vg, ag, lds, grid in rows:
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        w.writerow([n, grid, round(c / steps, 2), round(s / 1e6 / steps, 3), round(100 * s / tot, 2), round(a / 1e3, 1), round(mn / 1e3, 1), round(mx / 1e3, 1), vg, ag, lds])
print(f"{out}: {len(rows)} kernels, {tot / 1e6 / steps:.1f} ms of kernel time per step")
