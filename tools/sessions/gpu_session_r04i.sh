cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04i_f16_kernels.txt
PXA_OPERAND_DTYPE=f16 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "attention" 2>&1 | grep -v amdgpu | tail -40 > $o
PXA_OPERAND_DTYPE=f16 PXA_ATTN_NO_KVRES=1 timeout 900 python -m pytest tests/test_kernels_gpu.py -q -k "varlen_cross" 2>&1 | grep -v amdgpu | tail -5 >> $o
cat $o
