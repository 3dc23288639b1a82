#!/bin/bash
# round 4, session 39: the static item split as the persistent GEMMs' default (dynamic cursors switched on by the data-parallel runtime): GEMM kernel tests,
# the forced-collectives torchrun test (dynamic path), a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_39_static_default.txt
timeout 70 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "gemm and not nt4 and not k65536" > $F 2>&1; echo "gemm tests rc=$?" >> $F
timeout 60 python -m pytest tests/test_training_runtime_gpu.py -m gpu -q -x -p no:cacheprovider -k "forced_collectives" >> $F 2>&1; echo "torchrun forced collectives rc=$?" >> $F
echo "bench: $(timeout 60 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')" >> $F
grep -v amdgpu.ids $F | tail -8
