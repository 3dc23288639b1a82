# GPU-box session r03u: CAME step, per-launch times (before / after re-tiling the scalar-path tensors: PXA_CAME_OLD_TILES=1 restores one tile per 262,144 elements)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -x -q -m gpu -k "came" 2>&1 | tail -2 > gpurun_out/r03u_pytest.txt
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_came -o came -- python tools/bench_opt.py > gpurun_out/r03u_bench_opt.txt 2>&1
python tools/export_profile.py gpurun_out/prof_came/came_results.db gpurun_out/r03u_came_kernel_stats.csv 1
rm -rf gpurun_out/prof_came
cat gpurun_out/r03u_pytest.txt; grep parameters gpurun_out/r03u_bench_opt.txt; head -8 gpurun_out/r03u_came_kernel_stats.csv | cut -c1-200
