#!/bin/bash
# round 5 session 9: the counters VERDICT r04 item 2 asked for.  (a) stand-alone attention kernels, q prescaled (the step's instances) against the instances that
# multiply by scale log2 e themselves: instructions per launch by class; (b) the dK/dV kernel INSIDE bench.py's training step, 32-row default (dkv4) against the
# 80-row variant (dkv5): duration, effective clock, matrix-pipe busy share under rocprofv3 --pmc
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1 PXA_OPERAND_DTYPE=f16
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
F=$O/r5_09_pmc_attention_sq_prescale.txt
echo "$hdr; python tools/kbench.py attn under rocprofv3 --pmc, means per launch" > $F
for cfg in "q prescaled (the step's instances)|A=1" "q not prescaled (PXA_KBENCH_NO_PRESCALE=1: the <false> instances, round 4's operands)|PXA_KBENCH_NO_PRESCALE=1"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  echo "== $label" >> $F
  env $envs rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pq9 -o r -- python tools/kbench.py attn > /dev/null 2>&1
  python tools/pmc_query.py $O/pq9/r_results.db "attn_(fwd4|bwd_dq4|bwd_dkv4)" >> $F 2>&1
  rm -rf $O/pq9
done
G=$O/r5_09_pmc_step_dkv.txt
echo "$hdr; bench.py --steps 2 --warmup 1 under rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES; attention kernels of the step, means per launch over the two timed steps" > $G
for cfg in "dkv4 (default)|A=1" "dkv5 (PXA_ATTN_DKV=5)|PXA_ATTN_DKV=5"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  echo "== $label" >> $G
  env $envs timeout 420 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES -d $O/pq9s -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $O/pq9s.log 2>&1
  python tools/pmc_step_dump.py $O/pq9s/r_results.db $O/pq9s.csv "attn_(fwd4|bwd_dq4|bwd_dkv4|bwd_dkv5)" >> $O/pq9s.log 2>&1
  python tools/pmc_step_attn_summary.py $O/pq9s.csv >> $G 2>&1
  rm -rf $O/pq9s $O/pq9s.csv
done
cat $F | cut -c1-250; cat $G | cut -c1-250
