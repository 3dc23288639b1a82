#!/bin/bash
# round 4, session 38: two more round-2/3 GEMM choices decided in the step: the dynamic per-XCD item cursors (against the static split) and the half-width remainder items
# (against padding the remainder column to a full tile)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_38_step_ab_gemm_sched.txt
: > $F
run() { echo "$1: $(env $2 timeout 120 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')" >> $F; }
run "default" "A=1"
run "PXA_GEMM_STATIC=1" "PXA_GEMM_STATIC=1"
run "PXA_GEMM_NO_HALF_ITEMS=1" "PXA_GEMM_NO_HALF_ITEMS=1"
run "default" "A=1"
cat $F
