#!/bin/bash
# round 5 session 7 (re-entry): state of HEAD on one box - smoke, default bench line (all legs), step profile + HBM counters of the fp16 build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; tag=${1:-r5_07}
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown)"
echo "$hdr" > $O/${tag}_smoke.txt
timeout 600 python __graft_entry__.py smoke >> $O/${tag}_smoke.txt 2>&1; echo "smoke rc=$?" >> $O/${tag}_smoke.txt
timeout 900 python bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
timeout 600 bash tools/profile_round.sh $tag > $O/${tag}_profile_round.log 2>&1
{ echo "$hdr operand build f16"; python tools/family_times.py $O/${tag}_step_kernel_stats.csv; } > $O/${tag}_family_times.txt 2>&1
tail -n 3 $O/${tag}_smoke.txt; cut -c1-1500 $O/${tag}_bench_default.json; head -40 $O/${tag}_step_kernel_stats.csv | cut -c1-150; cat $O/${tag}_family_times.txt
