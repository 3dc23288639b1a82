#!/bin/bash
# round 5, session 6: cross_attn.proj bias gradient fused into ln_mod_bwd: parity, step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
for op in f16 bf16; do
  PXA_OPERAND_DTYPE=$op timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "ln_mod or cond_linear or gate" -p no:cacheprovider > $O/r5_06_pytest_rows_$op.txt 2>&1; echo "rc=$?" >> $O/r5_06_pytest_rows_$op.txt
done
PXA_OPERAND_DTYPE=f16 timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "training or grad" -p no:cacheprovider > $O/r5_06_pytest_model_f16.txt 2>&1; echo "rc=$?" >> $O/r5_06_pytest_model_f16.txt
F=$O/r5_06_step_ab.txt
echo "# box $(hostname) $(date -u +%FT%TZ) fp16 build, bench.py --steps 8 --warmup 3, two rounds" > $F
for rep in 1 2; do
for cfg in "default (bias gradient of cross_attn.proj in ln_mod_bwd)|A=1" "separate colsum pass|PXA_FUSED_CPROJ_BIAS=0"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  r=$(env $envs timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null \
      | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')
  echo "$label: $r" >> $F
done
done
for op in f16 bf16; do tail -3 $O/r5_06_pytest_rows_$op.txt; done; tail -3 $O/r5_06_pytest_model_f16.txt; cat $F
