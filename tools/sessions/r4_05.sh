#!/bin/bash
# round 4, session 5: attn_fwd4_kernel v3 (softmax split over both phases, K reads early, DMA in the second product's tail) - parity, time, ablations, SQ counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_fwd4.py all > $O/r4_05_fwd4_bf16.txt 2>&1; echo "bf16 rc=$?" >> $O/r4_05_fwd4_bf16.txt
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_fwd4.py all > $O/r4_05_fwd4_f16.txt 2>&1; echo "f16 rc=$?" >> $O/r4_05_fwd4_f16.txt
for v in fold nvq3 nvq4 nvq6 abl1 abl2 abl4 abl32 abl24 abl39 abl1f abl4f abl24f; do
  PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f4_$v.so timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=1" | tail -1 >> $O/r4_05_fwd4_variants.txt
done
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  PXA_OPERAND_DTYPE=f16 timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pq -o r -- python tools/kbench_fwd4.py time > /dev/null 2>&1
  echo "== f16: $ctr" >> $O/r4_05_pmc_fwd4_sq.txt
  python tools/pmc_query.py $O/pq/r_results.db "attn_fwd" >> $O/r4_05_pmc_fwd4_sq.txt 2>&1
  rm -rf $O/pq
done
grep -v amdgpu.ids $O/r4_05_fwd4_bf16.txt $O/r4_05_fwd4_f16.txt $O/r4_05_fwd4_variants.txt; cat $O/r4_05_pmc_fwd4_sq.txt
