#!/bin/bash
# round 4, session 22: SQ counters of the persistent GEMM (and of the vendor library's kernel on the same shape): MFMA busy, effective clock, instruction mix -
# the same "busy x clock" reading as profiles/r4_21_pmc_bwd_sq.txt gives for the attention kernels
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
CTR="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU"
P=$O/r4_22_pmc_gemm_sq.txt
: > $P
run() {   # label, pattern, env, args
  env $3 timeout 200 rocprofv3 --kernel-trace --pmc $CTR -d $O/pq -o r -- python tools/kbench_one.py $4 20 > $O/pq_stdout.txt 2>&1
  echo "== $1: kbench_one.py $4   [$(grep TF/s $O/pq_stdout.txt | tail -1)]" >> $P
  python tools/pmc_query.py $O/pq/r_results.db "$2" >> $P 2>&1
  rm -rf $O/pq $O/pq_stdout.txt
}
run "NT fc1 shape" gemm_pers "A=1" "65536 4608 1152 NT"
run "NT proj shape" gemm_pers "A=1" "65536 1152 1152 NT"
run "NT fc2 shape" gemm_pers "A=1" "65536 1152 4608 NT"
run "NN fc1 dX shape" gemm_pers "A=1" "65536 1152 4608 NN"
run "vendor library, NT fc1 shape" Cijk "KBENCH_LIBREF=1" "65536 4608 1152 NT"
cat $P
