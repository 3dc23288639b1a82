# GPU-box session r03v: config 2 (512px inference) per-kernel profile (60 NFE: three 20-step sampler runs) + eager vs HIP-graph replay wall time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_inf -o inf -- python tools/bench_infer.py 512 > gpurun_out/r03v_bench_infer_prof.txt 2>&1
python tools/export_profile.py gpurun_out/prof_inf/inf_results.db gpurun_out/r03v_infer512_kernel_stats.csv 60
rm -rf gpurun_out/prof_inf
timeout 300 python tools/bench_infer.py 512 2>&1 | grep workload > gpurun_out/r03v_bench_infer.txt
timeout 300 python tools/bench_infer.py 512 --graph 2>&1 | grep workload >> gpurun_out/r03v_bench_infer.txt
cat gpurun_out/r03v_bench_infer.txt | cut -c1-400; head -30 gpurun_out/r03v_infer512_kernel_stats.csv | cut -c1-170
