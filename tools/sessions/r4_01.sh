#!/bin/bash
# round 4, session 1: first run of attn_fwd4_kernel (parity, full grid, time, ablations) + test_kernels_gpu.py under both operand libraries
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_fwd4.py check > $O/r4_01_fwd4_check_bf16.txt 2>&1; echo "check bf16 rc=$?" >> $O/r4_01_fwd4_check_bf16.txt
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_fwd4.py check > $O/r4_01_fwd4_check_f16.txt 2>&1; echo "check f16 rc=$?" >> $O/r4_01_fwd4_check_f16.txt
timeout 120 python tools/kbench_fwd4.py time > $O/r4_01_fwd4_time.txt 2>&1
for v in dmaA abl1 abl2 abl4 abl32 abl8 abl16 abl24 abl39; do
  PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f4_$v.so timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=1" >> $O/r4_01_fwd4_time.txt
done
PXA_OPERAND_DTYPE=f16 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -s -p no:cacheprovider > $O/r4_01_kernels_f16.txt 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -p no:cacheprovider > $O/r4_01_kernels_bf16.txt 2>&1
tail -3 $O/r4_01_kernels_f16.txt $O/r4_01_kernels_bf16.txt
cat $O/r4_01_fwd4_check_bf16.txt $O/r4_01_fwd4_time.txt
