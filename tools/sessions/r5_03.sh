#!/bin/bash
# round 5, session 3: aux / bias requests issued before the loop-end barrier, prefetch under their latency: parity, kbench, step
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
V=pixart_sigma_amd/variants
timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" -p no:cacheprovider > $O/r5_03_pytest_gemm.txt 2>&1; echo "rc=$?" >> $O/r5_03_pytest_gemm.txt
timeout 900 python -m pytest tests/test_vae_gpu.py -q -x -p no:cacheprovider > $O/r5_03_pytest_vae.txt 2>&1; echo "rc=$?" >> $O/r5_03_pytest_vae.txt
export PXA_OPERAND_DTYPE=f16
echo "# box $(hostname) $(date -u +%FT%TZ) operand f16" > $O/r5_03_kbench_epi.txt
for v in "" f16_early0 ""; do
  if [ -z "$v" ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$V/lib_$v.so; fi
  timeout 120 python tools/kbench_epi.py 4 24 2>&1 | grep -v amdgpu >> $O/r5_03_kbench_epi.txt
done
unset PXA_LIB_PATH PXA_OPERAND_DTYPE
F=$O/r5_03_step_ab.txt
echo "# box $(hostname) $(date -u +%FT%TZ) fp16 build, bench.py --steps 8 --warmup 3, two rounds" > $F
for rep in 1 2; do
for cfg in "early (default)|A=1" "round-4 order|PXA_LIB_PATH=$V/lib_f16_early0.so"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  r=$(env $envs timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null \
      | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')
  echo "$label: $r" >> $F
done
done
tail -3 $O/r5_03_pytest_gemm.txt; tail -3 $O/r5_03_pytest_vae.txt; cat $O/r5_03_kbench_epi.txt; cat $F
