# PMC passes over one GEMM shape: ours (2-stage), ours (ping-pong), vendor library.  usage: bash tools/pmc_gemm.sh M N K LAYOUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  for v in base pp lib; do
    unset PXA_GEMM_PP KBENCH_LIBREF
    [ $v = pp ] && export PXA_GEMM_PP=1
    [ $v = lib ] && export KBENCH_LIBREF=1
    timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmcg_${v}_$i -o r -- python tools/kbench_one.py $1 $2 $3 $4 3 > gpurun_out/pmcg_${v}_$i.log 2>&1
    { grep TF gpurun_out/pmcg_${v}_$i.log; python tools/pmc_query.py gpurun_out/pmcg_${v}_$i/r_results.db "gemm|Cijk"; } > gpurun_out/pmcg_${v}_$i.txt 2>&1
    rm -rf gpurun_out/pmcg_${v}_$i gpurun_out/pmcg_${v}_$i.log
  done
done
