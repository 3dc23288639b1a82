#!/bin/bash
# round 4, session 20: prologue reorder (stationary rows behind the first DMA) + per-block slow-path skip: parity and time of all four one-wave kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
for dt in bf16 f16; do
  export PXA_OPERAND_DTYPE=$dt
  PXA_ATTN_FWD4=1 timeout 300 python tools/kbench_fwd4.py check 2>&1 | grep -v amdgpu.ids | grep -i "fail\|full grid" > $O/r4_20_$dt.txt
  KBENCH_QK_SCALE=1 timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=" | tail -2 >> $O/r4_20_$dt.txt
  KBENCH_QK_SCALE=2 timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=1" | tail -1 >> $O/r4_20_$dt.txt
  KB_DKV_MODE=5 timeout 300 python tools/kbench_dkv4.py all 2>&1 | grep -v amdgpu.ids | grep -i "fail\|full grid\|alone" >> $O/r4_20_$dt.txt
  timeout 300 python tools/kbench_dkv4.py dq 2>&1 | grep -v amdgpu.ids | grep -i "fail\|full grid\|alone" >> $O/r4_20_$dt.txt
done
cat $O/r4_20_bf16.txt $O/r4_20_f16.txt
