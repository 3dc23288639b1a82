# GPU-box session r03d: dK/dV kernel with MFMA-first slots + lean DMA issue: parity (attention tests) and timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $o/r03d_pytest_attention.log 2>&1
echo "pytest rc $?" >> $o/r03d_pytest_attention.log
for m in 0 2 2; do PXA_ATTN_DKV=$m timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dkv mode $m: /"; done > $o/r03d_attn_dkv_modes.txt
tail -3 $o/r03d_pytest_attention.log; cat $o/r03d_attn_dkv_modes.txt
