# GPU-box session r03x: static single-output GELU epilogue (EPI 7) - parity, kernel time, inference configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm" 2>&1 | tail -2 > gpurun_out/r03x_pytest.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "checkpoint or forward or sample or cfg or dpm" 2>&1 | tail -2 >> gpurun_out/r03x_pytest.txt
timeout 300 python tools/kbench_gelu.py 2>&1 | grep "M=" > gpurun_out/r03x_kbench_gelu.txt
timeout 300 python tools/bench_infer.py both 2>&1 | grep workload > gpurun_out/r03x_bench_infer.txt
cat gpurun_out/r03x_pytest.txt gpurun_out/r03x_kbench_gelu.txt; cut -c1-330 gpurun_out/r03x_bench_infer.txt
