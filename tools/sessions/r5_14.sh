#!/bin/bash
# round 5 session 14: the GPU test tier, smoke() and the default bench line once more at the round's last commit (kernels as in the r5final bundle; two more test cases)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; tag=${1:-r5head}
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown)"
echo "$hdr (both operand builds: the tier re-runs the kernel and model suites under f16 in subprocesses)" > $O/${tag}_pytest_gpu.txt
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider >> $O/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/${tag}_pytest_gpu.txt
echo "$hdr" > $O/${tag}_smoke.txt
timeout 900 python __graft_entry__.py smoke >> $O/${tag}_smoke.txt 2>&1; echo "smoke rc=$?" >> $O/${tag}_smoke.txt
timeout 1200 python bench.py > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
tail -n 3 $O/${tag}_pytest_gpu.txt; tail -n 3 $O/${tag}_smoke.txt | cut -c1-400; cut -c1-700 $O/${tag}_bench_default.json
