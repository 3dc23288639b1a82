#!/bin/bash
# round 4, session 7: full GPU test tier + default bench with attn_fwd4_kernel as the fp16 build's forward self-attention
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/r4_07_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/r4_07_pytest_gpu.txt
timeout 900 python bench.py > $O/r4_07_bench_default.json 2> $O/r4_07_bench_default.err
PXA_ATTN_FWD4=0 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-dtype --no-torch-baseline > $O/r4_07_bench_fwd2.json 2>> $O/r4_07_bench_default.err || true
tail -n 5 $O/r4_07_pytest_gpu.txt; cat $O/r4_07_bench_default.json | cut -c1-1500; tail -3 $O/r4_07_bench_default.err; cat $O/r4_07_bench_fwd2.json | cut -c1-600
