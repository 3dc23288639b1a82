# GPU-box session r03y: IDDPM ancestral sampler on the HIP path vs the reference golden (both operand builds) + scripts/inference.py with both samplers
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -q -s -k "iddpm" 2>&1 | grep "IDDPM\|passed\|failed" > gpurun_out/r03y_pytest.txt
PXA_OPERAND_DTYPE=f16 timeout 900 python -m pytest tests/test_model_gpu.py -q -s -k "iddpm" 2>&1 | grep "IDDPM\|passed\|failed" >> gpurun_out/r03y_pytest.txt
cat gpurun_out/r03y_pytest.txt
