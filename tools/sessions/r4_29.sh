#!/bin/bash
# round 4, session 29: what bounds the register-staged NT4 kernel (256 x 192 items, four-unit pipeline): no producer / no fragment reads / loads without their LDS writes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
A=$O/r4_29_nt4_ablations.txt
: > $A
KB_NT4_MODES=1 KB_NT4_SHAPES=fc1,fc2 timeout 200 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $A
for v in 1 2 32; do
  PXA_LIB_PATH=pixart_sigma_amd/variants/lib_nt4_abl$v.so KB_NT4_MODES=1 KB_NT4_SHAPES=fc1,fc2 timeout 200 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $A
done
cat $A
