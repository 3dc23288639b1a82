#!/bin/bash
# round 4, session 33: the 16-bit outputs of the row kernels (the NEXT GEMM's A operand) as plain stores instead of non-temporal ones - does the GEMM find them in the
# Infinity Cache?  Training step, fp16 build, same box, alternating: default library / variant (norm.hip with PXA_STREAM_NT=7)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_33_step_ab_keep16.txt
: > $F
for rep in 1 2; do
  for lib in default pixart_sigma_amd/variants/lib_keep16.so; do
    if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$lib; fi
    echo "lib=$lib: $(timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')" >> $F
  done
done
cat $F
