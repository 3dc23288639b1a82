#!/bin/bash
# round 4, session 31: does gemm_nt4_kernel pay in the training step?  Same box, alternating: PXA_GEMM_NT4 = 0 / default / 1 (every NT call it can take, fc2 included)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_31_step_ab_nt4.txt
: > $F
for rep in 1 2; do
  for mode in 0 default 1; do
    if [ $mode = default ]; then unset PXA_GEMM_NT4; else export PXA_GEMM_NT4=$mode; fi
    echo "PXA_GEMM_NT4=$mode: $(timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')" >> $F
  done
done
cat $F
