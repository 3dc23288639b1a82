#!/bin/bash
# round 4, session 13: attn_bwd_dkv4_kernel v2 (fragment pairs behind one wait, staged vector work) - parity, time, SQ counters vs attn_bwd_dkv2_kernel
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_dkv4.py all > $O/r4_13_dkv4_bf16.txt 2>&1; echo "bf16 rc=$?" >> $O/r4_13_dkv4_bf16.txt
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_dkv4.py all > $O/r4_13_dkv4_f16.txt 2>&1; echo "f16 rc=$?" >> $O/r4_13_dkv4_f16.txt
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM"; do
  PXA_OPERAND_DTYPE=f16 timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d $O/pq -o r -- python tools/kbench_dkv4.py time > /dev/null 2>&1
  echo "== f16: $ctr" >> $O/r4_13_pmc_dkv4_sq.txt
  python tools/pmc_query.py $O/pq/r_results.db "attn_bwd_dkv" >> $O/r4_13_pmc_dkv4_sq.txt 2>&1
  rm -rf $O/pq
done
grep -v amdgpu.ids $O/r4_13_dkv4_bf16.txt $O/r4_13_dkv4_f16.txt | grep -v " ok$"; cat $O/r4_13_pmc_dkv4_sq.txt
