# GPU-box session r03k: full GPU tier after the colsum inline-constant fix
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > gpurun_out/r03k_pytest_gpu.log 2>&1
echo "pytest rc $?" >> gpurun_out/r03k_pytest_gpu.log
grep -E "^FAILED" gpurun_out/r03k_pytest_gpu.log | head; tail -3 gpurun_out/r03k_pytest_gpu.log
