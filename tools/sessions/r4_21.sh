#!/bin/bash
# round 4, session 21: the three organisation A/Bs (two waves per SIMD vs one) with the box's power and shader clock sampled over every timed loop
# (tools/box_sampler.py) and one SQ pass each for the dK/dV and dQ pairs: MFMA busy, instruction mix, effective clock = GRBM_GUI_ACTIVE / wall
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
export PXA_OPERAND_DTYPE=f16
F=$O/r4_21_ab_power.txt
: > $F
timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=" >> $F
KB_DKV_MODE=5 timeout 120 python tools/kbench_dkv4.py time 2>&1 | grep "alone" >> $F
timeout 200 python tools/kbench_dkv4.py dq 2>&1 | grep "alone" >> $F
CTR="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VALU"
P=$O/r4_21_pmc_bwd_sq.txt
: > $P
KB_DKV_MODE=5 timeout 300 rocprofv3 --kernel-trace --pmc $CTR -d $O/pq -o r -- python tools/kbench_dkv4.py time > /dev/null 2>&1
echo "== f16, dK/dV pair: $CTR" >> $P; python tools/pmc_query.py $O/pq/r_results.db "attn_bwd_dkv" >> $P 2>&1; rm -rf $O/pq
timeout 300 rocprofv3 --kernel-trace --pmc $CTR -d $O/pq -o r -- python tools/kbench_dkv4.py dq > /dev/null 2>&1
echo "== f16, dQ pair: $CTR" >> $P; python tools/pmc_query.py $O/pq/r_results.db "attn_bwd_dq" >> $P 2>&1; rm -rf $O/pq
cat $F $P
