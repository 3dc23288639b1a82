#!/bin/bash
# round 4, session 27: gemm_nt4_kernel with register-staged line pairs: parity (both builds), time against the ping-pong kernel and the vendor library
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_27_nt4_regpairs.txt
timeout 300 python tools/kbench_nt4.py check > $F 2>&1; echo "bf16 check rc=$?" >> $F
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_nt4.py check >> $F 2>&1; echo "f16 check rc=$?" >> $F
timeout 300 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $F
grep -v amdgpu.ids $F | grep -v " ok$"
