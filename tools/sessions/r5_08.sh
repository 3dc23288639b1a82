#!/bin/bash
# round 5 session 8: ln_mod_bwd with the bias sums in LDS (the register version had slowed EVERY ln_mod_bwd call 204 -> 345 us, session 7 profile): parity,
# step A/B against the separate colsum pass, step profiles of the default and of the 80-row dK/dV kernel (why dkv5 loses in the step)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
for op in f16 bf16; do
  PXA_OPERAND_DTYPE=$op timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "ln_mod or gate" -p no:cacheprovider > $O/r5_08_pytest_rows_$op.txt 2>&1; echo "rc=$?" >> $O/r5_08_pytest_rows_$op.txt
done
F=$O/r5_08_step_ab.txt
bash tools/step_ab.sh $F.tmp "default (cross_attn.proj bias sums in ln_mod_bwd, LDS accumulators)|A=1" "separate colsum pass (round-4 ln_mod_bwd)|PXA_FUSED_CPROJ_BIAS=0" "dK/dV on the 80-row kernel (PXA_ATTN_DKV=5)|PXA_ATTN_DKV=5" > /dev/null 2>&1
{ echo "$hdr, bench.py --steps 8 --warmup 3, two rounds"; cat $F.tmp; } > $F; rm -f $F.tmp
for cfg in "dkv4|A=1" "dkv5|PXA_ATTN_DKV=5"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  env $envs rocprofv3 --kernel-trace --stats -d $O/prof_r5_08_$label -o step -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $O/prof_r5_08_$label.log 2>&1
  python tools/export_profile.py $O/prof_r5_08_$label/step_results.db $O/r5_08_step_kernel_stats_$label.csv 3
  rm -rf $O/prof_r5_08_$label
done
for op in f16 bf16; do tail -2 $O/r5_08_pytest_rows_$op.txt; done; cat $F
for l in dkv4 dkv5; do echo "== $l"; head -22 $O/r5_08_step_kernel_stats_$l.csv | cut -c1-110; done
