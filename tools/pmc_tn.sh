cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_tn -o r -- python tools/kbench.py gemm > /dev/null 2>&1
  python tools/pmc_query.py gpurun_out/pmc_tn/r_results.db "gemm_pers"
  rm -rf gpurun_out/pmc_tn
done
