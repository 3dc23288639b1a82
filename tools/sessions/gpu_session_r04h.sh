# GPU-box session r04h: cross-attention forward with all keys resident in LDS (attn_fwd_kvres_kernel): parity (both operand builds) and time vs the streaming kernel
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04h_kvres.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3 > $o
PXA_OPERAND_DTYPE=f16 timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "keys_resident or varlen" 2>&1 | tail -3 >> $o
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "forward or train" 2>&1 | tail -3 >> $o
for rep in 1 2; do
PXA_ATTN_NO_KVRES=1 timeout 300 python tools/kbench_cross.py 2>&1 | grep "L=" | sed "s/^/streaming: /" >> $o
timeout 300 python tools/kbench_cross.py 2>&1 | grep "L=" | sed "s/^/resident : /" >> $o
done
cat $o
