#!/bin/bash
# round 4, session 6: attn_fwd4_kernel placement A/B on one box (K reads spread / DMA in phase A / V prefetch depth), bf16 (no fold) and fold builds
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 120 python tools/kbench_fwd4.py time > $O/r4_06_fwd4_ab.txt 2>&1
for v in ks da ksda nvq4 ksnvq3 fold ksf ksdaf; do
  PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f4_$v.so timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=1" | tail -1 >> $O/r4_06_fwd4_ab.txt
done
timeout 120 python tools/kbench_fwd4.py time >> $O/r4_06_fwd4_ab.txt 2>&1
grep -v amdgpu.ids $O/r4_06_fwd4_ab.txt
