#!/bin/bash
# round 4, session 24: gemm_nt4_kernel after the no-bias fix: parity; SQ counters (MFMA busy, clock) of the fc1 / fc2 shapes; ablations (what bounds the main loop)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_nt4.py check > $O/r4_24_nt4_check.txt 2>&1; echo "bf16 check rc=$?" >> $O/r4_24_nt4_check.txt
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_nt4.py check >> $O/r4_24_nt4_check.txt 2>&1; echo "f16 check rc=$?" >> $O/r4_24_nt4_check.txt
A=$O/r4_24_nt4_ablations.txt
: > $A
KB_NT4_MODES=0,1,lib KB_NT4_SHAPES=fc1,fc2 timeout 200 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $A
for v in 1 2 3 4 7; do
  PXA_LIB_PATH=pixart_sigma_amd/variants/lib_nt4_abl$v.so KB_NT4_MODES=1 KB_NT4_SHAPES=fc1,fc2 timeout 200 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $A
done
CTR="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_INST_ANY"
P=$O/r4_24_pmc_nt4_sq.txt
: > $P
KB_NT4_MODES=1,lib KB_NT4_SHAPES=fc1,fc2 timeout 300 rocprofv3 --kernel-trace --pmc $CTR -d $O/pq -o r -- python tools/kbench_nt4.py time > /dev/null 2>&1
python tools/pmc_query.py $O/pq/r_results.db "gemm_nt4" >> $P 2>&1; python tools/pmc_query.py $O/pq/r_results.db "Cijk" >> $P 2>&1; rm -rf $O/pq
CTR2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS"
KB_NT4_MODES=1 KB_NT4_SHAPES=fc1,fc2 timeout 300 rocprofv3 --kernel-trace --pmc $CTR2 -d $O/pq -o r -- python tools/kbench_nt4.py time > /dev/null 2>&1
python tools/pmc_query.py $O/pq/r_results.db "gemm_nt4" >> $P 2>&1; rm -rf $O/pq
grep -v amdgpu.ids $O/r4_24_nt4_check.txt | grep -v " ok$"; cat $A; cat $P
