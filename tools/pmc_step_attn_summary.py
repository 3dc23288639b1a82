"""Mean per launch of the in-step counter dump (tools/pmc_step_dump.py) for each kernel name, warm-up step dropped by taking the last two thirds of the launches.
Usage: python tools/pmc_step_attn_summary.py dump.csv"""
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
per = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(dict)
for r in rows:
    k = r["kernel"].replace("void ", "").split("(")[0] + f" [grid {r['grid']}]"
    per[k][r["counter"]].append((int(r["order"]), float(r["value"])))
    dur[k][int(r["order"])] = float(r["dur_ns"]) / 1e3
for k, d in per.items():
    n = len(dur[k])
    keep = sorted(dur[k])[n // 3:]
    ks = set(keep)
    out = {c: sum(v for o, v in vs if o in ks) / max(1, sum(1 for o, _ in vs if o in ks)) for c, vs in d.items()}
    us = sum(dur[k][o] for o in keep) / len(keep)
    line = f"{k:60s} {len(keep):4d} launches  {us:8.1f} us"
    # conventions of tools/pmc_step_table.py: GRBM_GUI_ACTIVE sums the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES against 32 x SQ_BUSY_CYCLES
    if "GRBM_GUI_ACTIVE" in out and us > 0:
        line += f"  eff. clock {out['GRBM_GUI_ACTIVE'] / 8 / (us * 1e3):5.2f} GHz"
    if "SQ_VALU_MFMA_BUSY_CYCLES" in out and out.get("SQ_BUSY_CYCLES"):
        line += f"  MFMA busy {out['SQ_VALU_MFMA_BUSY_CYCLES'] / (32 * out['SQ_BUSY_CYCLES']):5.3f}"
    if "SQ_WAIT_ANY" in out and out.get("SQ_WAVE_CYCLES"):
        line += f"  wait {out['SQ_WAIT_ANY'] / out['SQ_WAVE_CYCLES']:4.2f}"
    print(line)
    print("      " + "  ".join(f"{c} {v:.4g}" for c, v in sorted(out.items())))
