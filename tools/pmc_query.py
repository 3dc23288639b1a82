"""Summarise a rocprofv3 --pmc sqlite database: per-kernel mean of each counter (+ launch geometry / registers / LDS).
Usage: python tools/pmc_query.py <results.db> [name-regex] [--json out.json]
--json merges the per-(kernel, grid) counter means into out.json ({"kernels": [{"kernel", "grid", "counters": {...}}]}): the file
bench.py reads roofline.traffic from (one --pmc pass per counter group, merged over calls)."""
import json, os, re, sqlite3, sys
jout = None
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    jout = sys.argv[i + 1]
    del sys.argv[i:i + 2]
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
if jout:
    doc = json.load(open(jout)) if os.path.exists(jout) else {"unit": "counter means per launch; FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them", "kernels": []}
    for k, d in by.items():
        if not pat.search(k):
            continue
        name, _, g = k.partition("  [grid ")
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        grid = int(g.rstrip("]")) // 256 if g else None          # workgroups (all these kernels run 256-thread workgroups) ... see wg below
        ent = next((e for e in doc["kernels"] if e["kernel"] == name and e.get("grid_threads") == (int(g.rstrip("]")) if g else None)), None)
        if ent is None:
            ent = {"kernel": name, "grid_threads": int(g.rstrip("]")) if g else None, "grid": grid, "counters": {}}
            doc["kernels"].append(ent)
        for c, (v, n) in d.items():
            ent["counters"][c] = v
    json.dump(doc, open(jout, "w"), indent=1)
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
