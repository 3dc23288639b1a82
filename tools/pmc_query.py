"""Summarise a rocprofv3 --pmc sqlite database: per-kernel mean of each counter."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "counters_collection" not in tabs:
    print("no counters_collection view; tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()])
    sys.exit(0)
cols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
rows = cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name").fetchall()
by = {}
for k, c, v, n in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", k)[:70]
    by.setdefault(k, {})[c] = (v, n)
for k, d in by.items():
    if not any(s in k for s in ("gemm", "attn", "ln_mod", "gate", "colsum")):
        continue
    print(k)
    for c, (v, n) in sorted(d.items()):
        print(f"    {c:32s} {v:16.1f}  (n={n})")
