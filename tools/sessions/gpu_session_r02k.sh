# GPU-box session: closing evidence of the round after the streaming-access change: default bench line (both baselines) + rocprofv3 step profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 170 python bench.py > $o/r02k_bench_default.json 2> $o/r02k_bench_default.err
rocprofv3 --kernel-trace --stats -d $o/prof_r02k -o step -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline > $o/prof_r02k_step.log 2>&1
python tools/export_profile.py $o/prof_r02k/step_results.db $o/r02k_step_kernel_stats.csv 3
rm -rf $o/prof_r02k
cut -c1-400 $o/r02k_bench_default.json
