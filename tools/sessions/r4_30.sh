#!/bin/bash
# round 4, session 30: gemm_nt4_kernel, LDS-DMA line pairs issued behind an early-read barrier (2.6 units of look-ahead): parity, time; the first form (NT4_PAIRS=0) on the same box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_30_nt4_pairs_midbarrier.txt
timeout 300 python tools/kbench_nt4.py check > $F 2>&1; echo "bf16 check rc=$?" >> $F
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_nt4.py check >> $F 2>&1; echo "f16 check rc=$?" >> $F
KB_NT4_MODES=0,1,1,lib timeout 300 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $F
PXA_LIB_PATH=pixart_sigma_amd/variants/lib_nt4_singles.so KB_NT4_MODES=1,1 timeout 300 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $F
grep -v amdgpu.ids $F | grep -v " ok$"
