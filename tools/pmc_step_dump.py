"""Per-DISPATCH dump of a rocprofv3 --pmc (+ --kernel-trace) sqlite database taken over bench.py: one CSV row per (dispatch, counter) for kernels matching
a regex, in dispatch order, so that the launches of one kernel instance can be told apart by their position in the training step's fixed sequence
(tools/pmc_step_table.py labels them).  Usage: python tools/pmc_step_dump.py <results.db> <out.csv> [name-regex]"""
import csv, re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
out = sys.argv[2]
pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"gemm|splitk")
cur = db.cursor()
ccols = [d[1] for d in cur.execute("pragma table_info(counters_collection)")]
print("counters_collection columns:", ccols)
order = next((c for c in ("dispatch_id", "start", "id", "event_id") if c in ccols), None)
gcol = next((c for c in ccols if c.lower() in ("grid_size", "grid_size_x", "grid_x")), None)
dur = "end - start" if "start" in ccols and "end" in ccols else "0"
sel = f"select {order or 'rowid'}, kernel_name, {gcol or '0'}, counter_name, value, {dur} from counters_collection order by 1"
n = 0
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["order", "kernel", "grid", "counter", "value", "dur_ns"])
    for o, k, g, c, v, d in cur.execute(sel):
        if pat.search(k):
            w.writerow([o, re.sub(r"\(anonymous namespace\)::", "", k), g, c, v, d])
            n += 1
print(f"{out}: {n} rows")
