#!/bin/bash
# round 4, session 17: ablations of attn_bwd_dkv4_kernel (fp16 build): which instruction class costs what
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1 PXA_OPERAND_DTYPE=f16
timeout 120 python tools/kbench_dkv4.py time 2>&1 | grep "DKV=4" | tail -1 > $O/r4_17_dkv4_ablations.txt
for v in a1 a7 a8 a15 a16 a24 a31 a32; do
  PXA_LIB_PATH=pixart_sigma_amd/variants/lib_dk_$v.so timeout 120 python tools/kbench_dkv4.py time 2>&1 | grep "DKV=4" | tail -1 >> $O/r4_17_dkv4_ablations.txt
done
cat $O/r4_17_dkv4_ablations.txt
