# GPU-box session r03s: validation of the round's HEAD: full GPU tier, smoke(), default bench line, step profile + PMC, kbench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $o/r03s_pytest_gpu_tail.txt
timeout 900 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -4 > $o/r03s_smoke.txt
timeout 900 python bench.py > $o/r03s_bench_default.json 2> $o/r03s_bench_default.err
timeout 900 bash tools/profile_round.sh r03s > /dev/null 2>&1
KBENCH_LIBREF=1 timeout 600 python tools/kbench.py all > $o/r03s_kbench.txt 2>&1
timeout 300 python tools/kbench_elem.py 2>&1 | grep -v amdgpu.ids > $o/r03s_kbench_elem.txt
cat $o/r03s_pytest_gpu_tail.txt $o/r03s_smoke.txt; cut -c1-1800 $o/r03s_bench_default.json; echo; head -24 $o/r03s_step_kernel_stats.csv | cut -c1-150
