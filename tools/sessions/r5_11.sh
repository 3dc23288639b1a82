#!/bin/bash
# round 5 session 11: where the fp16-operand build (the timed one) spends the 10 ms it takes over the bf16 build on the same box: kernel traces of both builds' steps
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown)"
for dt in fp16 bf16 fp16 bf16; do
  python bench.py --dtype $dt --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["dtype"], d["ms_per_step"])' >> $O/r5_11_step_dtype.tmp
done
{ echo "$hdr; bench.py --dtype X --steps 8 --warmup 3, alternating"; cat $O/r5_11_step_dtype.tmp; } > $O/r5_11_step_dtype.txt; rm -f $O/r5_11_step_dtype.tmp
for dt in fp16 bf16; do
  rocprofv3 --kernel-trace --stats -d $O/prof_r5_11_$dt -o step -- python bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $O/prof_r5_11_$dt.log 2>&1
  python tools/export_profile.py $O/prof_r5_11_$dt/step_results.db $O/r5_11_step_kernel_stats_$dt.csv 3
  rm -rf $O/prof_r5_11_$dt
done
cat $O/r5_11_step_dtype.txt
for dt in fp16 bf16; do python tools/family_times.py $O/r5_11_step_kernel_stats_$dt.csv; done
