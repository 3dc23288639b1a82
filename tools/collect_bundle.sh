#!/bin/bash
# After a GPU session (tools/gpurun_session.sh TIMEOUT TAG recipe ...): copy what the judge reads from gpurun_out/ (scratch) into profiles/ (tracked) - every
# TAG_* text / csv / json artefact except raw logs - plus the session's parity summaries when they come from a full tier run.
#   usage (this container, repo root):  bash tools/collect_bundle.sh TAG
tag=${1:?tag}
cd "$(dirname "$0")/.." || exit 1
for f in gpurun_out/${tag}_*; do
  case "$f" in *.log|*.err|*_trace.csv|*_pmc_step_[0-9].csv) continue;; esac
  [ -f "$f" ] && [ "$(stat -c %s "$f")" -lt 2000000 ] && cp "$f" profiles/
done
for op in f16 bf16; do
  python - <<PY
import json, shutil, os
p = "gpurun_out/parity_summary_$op.json"
if os.path.exists(p) and len(json.load(open(p)).get("entries", [])) >= 50:
    shutil.copy(p, "profiles/${tag}_parity_summary_$op.json")
PY
done
ls profiles/${tag}_* 2>/dev/null
