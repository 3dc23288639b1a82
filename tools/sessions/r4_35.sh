#!/bin/bash
# round 4, session 35: HEAD with the dK/dV default back on attn_bwd_dkv4_kernel (session 34): smoke, the attention and model GPU tests, the default bench
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 150 python __graft_entry__.py smoke > $O/r4final_smoke.txt 2>&1; echo "smoke rc=$?" >> $O/r4final_smoke.txt
timeout 200 python bench.py > $O/r4final_bench_default.json 2> $O/r4final_bench_default.err
timeout 110 python -m pytest tests/test_model_gpu.py tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "attention or train or golden or grad or block" > $O/r4_35_pytest_subset.txt 2>&1; echo "pytest rc=$?" >> $O/r4_35_pytest_subset.txt
tail -n 2 $O/r4final_smoke.txt | cut -c1-300; cut -c1-400 $O/r4final_bench_default.json; tail -n 3 $O/r4_35_pytest_subset.txt
