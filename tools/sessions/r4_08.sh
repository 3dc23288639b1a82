#!/bin/bash
# round 4, session 8: step kernel profile (fp16 build, attn_fwd4 on) + smoke with parity lines
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
export PYTHONUNBUFFERED=1
rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $O/r4_08_prof_step.log 2>&1
python tools/export_profile.py $O/prof/step_results.db $O/r4_08_step_kernel_stats.csv 3
rm -rf $O/prof
timeout 600 python __graft_entry__.py smoke > $O/r4_08_smoke.txt 2>&1
tail -3 $O/r4_08_smoke.txt; head -40 $O/r4_08_step_kernel_stats.csv | cut -c1-150
