#!/bin/bash
# round 4, session 18: validation bundle - GPU test tier, smoke, default bench (all legs), round profile (step kernel stats + HBM counters), inference configs
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r4final_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/r4final_pytest_gpu.txt
timeout 900 python __graft_entry__.py smoke > $O/r4final_smoke.txt 2>&1; echo "smoke rc=$?" >> $O/r4final_smoke.txt
timeout 1200 python bench.py > $O/r4final_bench_default.json 2> $O/r4final_bench_default.err
timeout 900 bash tools/profile_round.sh r4final > $O/r4final_profile_round.log 2>&1
timeout 600 python tools/bench_infer.py > $O/r4final_bench_infer.txt 2>&1
timeout 600 python tools/bench_dmd.py >> $O/r4final_bench_infer.txt 2>&1
tail -n 3 $O/r4final_pytest_gpu.txt; tail -n 3 $O/r4final_smoke.txt; cut -c1-700 $O/r4final_bench_default.json; head -12 $O/r4final_step_kernel_stats.csv | cut -c1-120; grep -v amdgpu $O/r4final_bench_infer.txt | tail -8
