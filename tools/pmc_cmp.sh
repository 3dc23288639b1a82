# PMC comparison: the schedule probe (MODE 3/4 of probe/dma_bench) vs the product GEMM on the same shape
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pc_probe_$i -o r -- ./probe/dma_bench 3 s 1 > /dev/null 2>&1
  python tools/pmc_query.py gpurun_out/pc_probe_$i/r_results.db "sched_kernel" > gpurun_out/pc_probe_$i.txt 2>&1
  KBENCH_DATA=u timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pc_real_$i -o r -- python tools/kbench_one.py 8192 8192 8192 NT 3 > /dev/null 2>&1
  python tools/pmc_query.py gpurun_out/pc_real_$i/r_results.db "gemm" > gpurun_out/pc_real_$i.txt 2>&1
  rm -rf gpurun_out/pc_probe_$i gpurun_out/pc_real_$i
done
