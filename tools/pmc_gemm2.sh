# PMC passes over one GEMM shape: ours vs the vendor library (yardstick only).  usage: bash tools/pmc_gemm2.sh M N K LAYOUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  for v in base lib; do
    unset KBENCH_LIBREF
    [ $v = lib ] && export KBENCH_LIBREF=1
    timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmcg_${v}_$i -o r -- python tools/kbench_one.py $1 $2 $3 $4 20 > gpurun_out/pmcg_${v}_$i.log 2>&1
    { echo "== $v: $ctr"; grep TF gpurun_out/pmcg_${v}_$i.log; python tools/pmc_query.py gpurun_out/pmcg_${v}_$i/r_results.db "gemm_pers|Cijk"; }
    rm -rf gpurun_out/pmcg_${v}_$i gpurun_out/pmcg_${v}_$i.log
  done
done
