#!/bin/bash
# round 4, session 4: SQ counters of attn_fwd4_kernel vs attn_fwd2_kernel (separate --pmc passes, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
export PYTHONUNBUFFERED=1
for dt in f16 bf16; do
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT"; do
  PXA_OPERAND_DTYPE=$dt timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pq -o r -- python tools/kbench_fwd4.py time > /dev/null 2>&1
  echo "== $dt: $ctr" >> $O/r4_04_pmc_fwd4_sq.txt
  python tools/pmc_query.py $O/pq/r_results.db "attn_fwd" >> $O/r4_04_pmc_fwd4_sq.txt 2>&1
  rm -rf $O/pq
done
done
cat $O/r4_04_pmc_fwd4_sq.txt
