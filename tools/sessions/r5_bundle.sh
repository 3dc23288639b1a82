#!/bin/bash
# round 5 validation bundle (tag $1, default r5final): GPU test tier, smoke, default bench (all legs), round profile (step kernel stats + HBM counters, fp16 build),
# in-step GEMM counters, inference configs.  Every text artefact starts with the box, the time, the HEAD and the operand build.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
tag=${1:-r5final}
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown)"
echo "$hdr (both operand builds: the tier re-runs the kernel and model suites under f16 in subprocesses)" > $O/${tag}_pytest_gpu.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider >> $O/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/${tag}_pytest_gpu.txt
echo "$hdr" > $O/${tag}_smoke.txt
timeout 900 python __graft_entry__.py smoke >> $O/${tag}_smoke.txt 2>&1; echo "smoke rc=$?" >> $O/${tag}_smoke.txt
timeout 1200 python bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
timeout 900 bash tools/profile_round.sh $tag > $O/${tag}_profile_round.log 2>&1
timeout 900 bash tools/pmc_step.sh $tag > $O/${tag}_pmc_step.log 2>&1
{ echo "$hdr operand build f16"; python tools/pmc_step_table.py $O/$tag; } > $O/${tag}_pmc_step_gemm_table.txt 2>&1
echo "$hdr operand build f16 (tools/bench_infer.py, tools/bench_dmd.py)" > $O/${tag}_bench_infer.txt
timeout 600 python tools/bench_infer.py >> $O/${tag}_bench_infer.txt 2>&1
timeout 600 python tools/bench_dmd.py >> $O/${tag}_bench_infer.txt 2>&1
tail -n 3 $O/${tag}_pytest_gpu.txt; tail -n 3 $O/${tag}_smoke.txt; cut -c1-900 $O/${tag}_bench_default.json; head -14 $O/${tag}_step_kernel_stats.csv | cut -c1-130; cat $O/${tag}_pmc_step_gemm_table.txt | cut -c1-150; grep -v amdgpu $O/${tag}_bench_infer.txt | tail -5 | cut -c1-400
