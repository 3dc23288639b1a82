# GPU-box session: attention kernel parity after removing the losing variants from csrc/attn.hip
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 80 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > gpurun_out/r02m_pytest_attention.log 2>&1; echo "pytest rc $?" >> gpurun_out/r02m_pytest_attention.log
tail -3 gpurun_out/r02m_pytest_attention.log
