# GPU-box session r03p: ping-pong dK/dV kernel, prefetch distance of the matrix phase 2 / 3 / 4 / 6 fragments
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
for d in 2 3 4 6; do PXA_ATTN_DKV3_DEPTH=$d PXA_ATTN_DKV=3 timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "dkv_kernel_modes and 3-1" 2>&1 | tail -1 | sed "s/^/depth $d: /"; done > $o/r03p_pytest.txt
for d in 2 3 4 6; do PXA_ATTN_DKV3_DEPTH=$d PXA_ATTN_DKV=3 timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dkv3 depth $d: /"; done > $o/r03p_dkv3_depth.txt
PXA_ATTN_DKV=2 timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dkv mode 2:   /" >> $o/r03p_dkv3_depth.txt
cat $o/r03p_pytest.txt $o/r03p_dkv3_depth.txt
