#!/bin/bash
# round 4, session 10: does the forward attention's in-step slowdown come from the score range (deferred-maximum events)?  time vs q/k scale, threshold 6 vs 11
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1 PXA_OPERAND_DTYPE=f16
for qs in 1 1.5 2 3; do
  for v in th6 th11; do
    KBENCH_QK_SCALE=$qs PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f4_$v.so timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=" | tail -2 >> $O/r4_10_fwd_vs_score_range.txt
  done
done
KBENCH_QK_SCALE=3 PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f4_th11.so timeout 300 python tools/kbench_fwd4.py check 2>&1 | grep -v amdgpu >> $O/r4_10_fwd_vs_score_range.txt
cat $O/r4_10_fwd_vs_score_range.txt
