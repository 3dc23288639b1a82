#!/bin/bash
# round 4, session 37: the GEMMs' MFMA shape decided IN the step (the 16-row shape won the warm stand-alone bench by 8-11 % in round 2; the attention dK/dV kernel's 16-row
# variant won alone and lost in the step, session 34): default library (NT / NN on 16 x 16 x 32) against gemm.hip built with GEMM_M16=0 (everything on 32 x 32 x 16)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_37_step_ab_gemm_shape.txt
: > $F
for rep in 1 2; do
  for lib in default pixart_sigma_amd/variants/lib_gemm_m32.so; do
    if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$lib; fi
    echo "lib=$lib: $(timeout 200 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')" >> $F
  done
done
cat $F
