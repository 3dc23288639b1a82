# GPU-box session: smoke + model-level parity on the final code (the full GPU tier last ran before the streaming-access change)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 python __graft_entry__.py smoke > gpurun_out/r02l_smoke.log 2>&1; echo "smoke rc $?" >> gpurun_out/r02l_smoke.log
timeout 95 python -m pytest tests/test_model_gpu.py -x -q > gpurun_out/r02l_pytest_model.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02l_pytest_model.log
grep -v amdgpu gpurun_out/r02l_smoke.log | tail -3; tail -4 gpurun_out/r02l_pytest_model.log
