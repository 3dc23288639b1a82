# GPU-box session r03o: dK/dV as an 8-wave phase ping-pong (mode 3): parity, all-heads determinism check, timing vs mode 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $o/r03o_pytest_attention.log 2>&1
echo "pytest rc $?" >> $o/r03o_pytest_attention.log
timeout 600 python tools/dbg_attn_r03.py grid 2>&1 | grep -v amdgpu.ids > $o/r03o_dbg_grid.txt
for m in 3 2 3 2; do PXA_ATTN_DKV=$m timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dkv mode $m: /"; done > $o/r03o_attn_dkv_modes.txt
for m in 3 2; do PXA_ATTN_DKV=$m timeout 300 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids | grep cross | sed "s/^/dkv mode $m: /"; done >> $o/r03o_attn_dkv_modes.txt
tail -4 $o/r03o_pytest_attention.log; cat $o/r03o_dbg_grid.txt $o/r03o_attn_dkv_modes.txt
