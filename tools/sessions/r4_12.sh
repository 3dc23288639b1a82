#!/bin/bash
# round 4, session 12: first run of attn_bwd_dkv4_kernel (one wave per SIMD, 64 keys per wave): parity both builds, full grid, time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_dkv4.py all > $O/r4_12_dkv4_bf16.txt 2>&1; echo "bf16 rc=$?" >> $O/r4_12_dkv4_bf16.txt
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_dkv4.py all > $O/r4_12_dkv4_f16.txt 2>&1; echo "f16 rc=$?" >> $O/r4_12_dkv4_f16.txt
grep -v amdgpu.ids $O/r4_12_dkv4_bf16.txt $O/r4_12_dkv4_f16.txt
