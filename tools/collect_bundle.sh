#!/bin/bash
# After `gpurun -- 'bash tools/sessions/r5_bundle.sh [tag]'` (the validation bundle, default tag r5final): copy what the judge reads from gpurun_out/ (scratch) into profiles/ (tracked).
#   usage (this container, repo root):  bash tools/collect_bundle.sh [tag]
tag=${1:-r5final}
cd "$(dirname "$0")/.." || exit 1
for f in pytest_gpu.txt smoke.txt bench_default.json step_kernel_stats.csv pmc_attention.txt pmc_attention.json pmc_gemm.txt bench_infer.txt profile_round.log pmc_step_gemm_table.txt; do
  [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
# (the per-session parity summaries are overwritten by EVERY GPU pytest session: take them only when they are the full tier's)
for op in f16 bf16; do
  python - <<PY
import json, shutil, os
p = "gpurun_out/parity_summary_$op.json"
if os.path.exists(p) and len(json.load(open(p)).get("entries", [])) >= 50:
    shutil.copy(p, "profiles/${tag}_parity_summary_$op.json")
PY
done
python tools/family_times.py profiles/${tag}_step_kernel_stats.csv
python - <<PY
import json
d = json.loads(open("profiles/${tag}_bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print(f"{d['ms_per_step']:.1f} ms/step = {d['value']:.3f} steps/s ({d['dtype']}); other dtype {d.get('other_dtype', {}).get('ms_per_step')}; dominant {r['kernel'].split()[0]} {r['ms_per_launch']:.3f} ms frac {r['frac']:.3f} traffic {r['traffic']}")
PY
