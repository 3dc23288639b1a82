# GPU-box session r03w: single-output GELU for forwards whose GELU' nobody reads (inference, the discarded forward of a checkpointed step)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -x -q -k "epilogues or checkpoint or forward or sample or cfg or dpm" 2>&1 | tail -3 > gpurun_out/r03w_pytest.txt
PXA_OPERAND_DTYPE=f16 timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "checkpoint or forward or sample or cfg or dpm" 2>&1 | tail -3 >> gpurun_out/r03w_pytest.txt
timeout 300 python tools/bench_infer.py both 2>&1 | grep workload > gpurun_out/r03w_bench_infer.txt
timeout 300 python bench.py --grad-checkpoint --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype 2>&1 | tail -1 | cut -c1-400 > gpurun_out/r03w_bench_gradckpt.txt
cat gpurun_out/r03w_pytest.txt; cut -c1-330 gpurun_out/r03w_bench_infer.txt; cat gpurun_out/r03w_bench_gradckpt.txt
