#!/bin/bash
# round 5, session 5: lse folded into the dQ kernel's first product (prescaled instance), conditioning linears on HIP: parity on both builds, step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
V=pixart_sigma_amd/variants
for op in f16 bf16; do
  PXA_OPERAND_DTYPE=$op timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "prescaled or attention or cond_linear" -p no:cacheprovider > $O/r5_05_pytest_attn_$op.txt 2>&1; echo "rc=$?" >> $O/r5_05_pytest_attn_$op.txt
  PXA_OPERAND_DTYPE=$op timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -p no:cacheprovider > $O/r5_05_pytest_model_$op.txt 2>&1; echo "rc=$?" >> $O/r5_05_pytest_model_$op.txt
done
F=$O/r5_05_step_ab.txt
echo "# box $(hostname) $(date -u +%FT%TZ) fp16 build, bench.py --steps 8 --warmup 3, two rounds" > $F
for rep in 1 2; do
for cfg in "default (lse folded in dq4)|A=1" "dq4 without the lse fold|PXA_LIB_PATH=$V/lib_f16_dq4nofold.so" "PXA_Q_PRESCALE=0|PXA_Q_PRESCALE=0"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  r=$(env $envs timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null \
      | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')
  echo "$label: $r" >> $F
done
done
{ echo "# bench.py per-kernel legs (prescaled operands), default library then the no-fold variant"; 
  timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-dtype --no-torch-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d["roofline"]["kernels"]))';
  PXA_LIB_PATH=$V/lib_f16_dq4nofold.so timeout 300 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-dtype --no-torch-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps(d["roofline"]["kernels"]))'; } > $O/r5_05_kernel_legs.txt 2>&1
for op in f16 bf16; do tail -3 $O/r5_05_pytest_attn_$op.txt; tail -3 $O/r5_05_pytest_model_$op.txt; done
cat $F; cat $O/r5_05_kernel_legs.txt
