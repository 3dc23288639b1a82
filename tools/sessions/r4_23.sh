#!/bin/bash
# round 4, session 23: gemm_nt4_kernel (one wave per SIMD NT GEMM) - first run: parity against fp32 and the ping-pong kernel, then the A/B at the step's shapes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
timeout 300 python tools/kbench_nt4.py check > $O/r4_23_nt4_bf16.txt 2>&1; echo "check rc=$?" >> $O/r4_23_nt4_bf16.txt
timeout 300 python tools/kbench_nt4.py time >> $O/r4_23_nt4_bf16.txt 2>&1
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_nt4.py check > $O/r4_23_nt4_f16.txt 2>&1; echo "check rc=$?" >> $O/r4_23_nt4_f16.txt
grep -v amdgpu.ids $O/r4_23_nt4_bf16.txt $O/r4_23_nt4_f16.txt
