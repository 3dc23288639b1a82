cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pq -o r -- python tools/kbench.py attn > /dev/null 2>&1
  python tools/pmc_query.py gpurun_out/pq/r_results.db "attn_(fwd|bwd)"
  rm -rf gpurun_out/pq
done
