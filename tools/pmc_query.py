"""Summarise a rocprofv3 --pmc sqlite database: per-kernel mean of each counter (+ launch geometry / registers / LDS).
Usage: python tools/pmc_query.py <results.db> [name-regex]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else r"gemm|attn|ln_mod|gate|colsum|Cijk")
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "counters_collection" not in tabs:
    print("no counters_collection view; tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()])
    sys.exit(0)
ccols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
gcol = next((c for c in ccols if c.lower() in ("grid_size", "grid_size_x", "grid_x")), None)      # separate launches of different sizes
if gcol:
    rows = cur.execute(f"select kernel_name || '  [grid ' || {gcol} || ']', counter_name, avg(value), count(*) from counters_collection group by kernel_name, {gcol}, counter_name").fetchall()
else:
    print("counters_collection columns:", ccols)
    rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
info = {}
try:
    kcols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    wg = next((c for c in kcols if c.lower() in ("workgroup_size_x", "workgroup_size", "workgroup_x")), "0")
    gr = next((c for c in kcols if c.lower() in ("grid_size_x", "grid_size", "grid_x")), "0")
    for r in cur.execute(f"select name, avg(end-start), max(vgpr_count), max(accum_vgpr_count), max(lds_size), max({wg}), max({gr}), count(*) from kernels group by name"):
        info[r[0]] = r[1:]
except sqlite3.Error as e:
    print("kernel info unavailable:", e)
by = {}
for k, c, v, n in rows:
    by.setdefault(k, {})[c] = (v, n)
for k, d in by.items():
    if not pat.search(k):
        continue
    print(re.sub(r"\(anonymous namespace\)::", "", k)[:160])
    kb = k.split("  [grid ")[0]
    if kb in info:
        i = info[kb]
        print(f"    avg_us={i[0]/1e3:.1f} vgpr={i[1]} agpr={i[2]} lds={i[3]} wg={i[4]} grid={i[5]} launches={i[6]}")
    for c, (v, n) in sorted(d.items()):
        print(f"    {c:32s} {v:16.1f}  (n={n})")
