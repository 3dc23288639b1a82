# GPU-box session r04f: final validation of the round's HEAD: full GPU tier, smoke(), default bench line, step profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $o/r03final_pytest_gpu_tail.txt
timeout 900 python __graft_entry__.py smoke 2>&1 | grep -v amdgpu.ids | tail -3 > $o/r03final_smoke.txt
timeout 900 python bench.py > $o/r03final_bench_default.json 2> $o/r03final_bench_default.err
rocprofv3 --kernel-trace --stats -d $o/prof_final -o step -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $o/prof_final_step.log 2>&1
python tools/export_profile.py $o/prof_final/step_results.db $o/r03final_step_kernel_stats.csv 3
rm -rf $o/prof_final
timeout 300 python bench.py --optimizer came --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype 2>&1 | tail -1 | cut -c1-600 > $o/r03final_bench_came.txt
cat $o/r03final_pytest_gpu_tail.txt $o/r03final_smoke.txt; cut -c1-700 $o/r03final_bench_default.json; echo; cat $o/r03final_bench_came.txt | cut -c1-420; head -22 $o/r03final_step_kernel_stats.csv | cut -c1-140
