#!/bin/bash
# round 4, session 15: full GPU tier + default bench with attn_bwd_dkv4_kernel as the default dK/dV kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/r4_15_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/r4_15_pytest_gpu.txt
timeout 900 python bench.py --no-cpu-baseline --no-torch-baseline > $O/r4_15_bench.json 2> $O/r4_15_bench.err
tail -n 4 $O/r4_15_pytest_gpu.txt; cut -c1-900 $O/r4_15_bench.json; tail -2 $O/r4_15_bench.err
