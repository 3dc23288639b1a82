#!/bin/bash
# round 4, session 34: do the attention kernels' stand-alone gains arrive in the training step?  fp16 build, same box, two rounds of: default / dK/dV on dkv4 / on the
# two-wave dkv2 / forward on the two-wave fwd2 / dQ on the two-wave dq2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_34_step_ab_attention.txt
: > $F
run() { echo "$1: $(env $2 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')" >> $F; }
for rep in 1 2; do
  run "default (fwd4, dq4, dkv5)" "A=1"
  run "PXA_ATTN_DKV=4" "PXA_ATTN_DKV=4"
  run "PXA_ATTN_DKV=2" "PXA_ATTN_DKV=2"
  run "PXA_ATTN_FWD4=0" "PXA_ATTN_FWD4=0"
  run "PXA_ATTN_DQ=1" "PXA_ATTN_DQ=1"
done
cat $F
