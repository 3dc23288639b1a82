#!/bin/bash
# round 5, session 4: softmax scale folded into the qkv projection (q_prescaled): kernel + model parity on both builds, step A/B; VAE after the aux-epilogue change; full bench line
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
for op in f16 bf16; do
  PXA_OPERAND_DTYPE=$op timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "prescaled or scale_copy or attention" -p no:cacheprovider > $O/r5_04_pytest_attn_$op.txt 2>&1; echo "rc=$?" >> $O/r5_04_pytest_attn_$op.txt
  PXA_OPERAND_DTYPE=$op timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -p no:cacheprovider > $O/r5_04_pytest_model_$op.txt 2>&1; echo "rc=$?" >> $O/r5_04_pytest_model_$op.txt
done
F=$O/r5_04_step_ab.txt
echo "# box $(hostname) $(date -u +%FT%TZ) fp16 build, bench.py --steps 8 --warmup 3, two rounds" > $F
for rep in 1 2; do
for cfg in "q prescaled (default)|A=1" "PXA_Q_PRESCALE=0|PXA_Q_PRESCALE=0"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  r=$(env $envs timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null \
      | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')
  echo "$label: $r" >> $F
done
done
timeout 600 python bench.py --no-cpu-baseline --no-torch-baseline --no-other-dtype > $O/r5_04_bench_roofline.json 2> $O/r5_04_bench_roofline.err
timeout 600 python tools/bench_infer.py > $O/r5_04_bench_infer.txt 2>&1
timeout 600 python tools/bench_dmd.py >> $O/r5_04_bench_infer.txt 2>&1
for op in f16 bf16; do tail -4 $O/r5_04_pytest_attn_$op.txt; tail -4 $O/r5_04_pytest_model_$op.txt; done
cat $F; cut -c1-1500 $O/r5_04_bench_roofline.json; grep -v amdgpu $O/r5_04_bench_infer.txt | tail -6
