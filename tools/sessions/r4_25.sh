#!/bin/bash
# round 4, session 25: is the NT4 kernel's DMA stream bound by half-line (64-byte row segment) requests?  Same bytes fetched as whole 128-byte lines (NT4_ABL=8, wrong data)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
A=$O/r4_25b_nt4_linehalves.txt
: > $A
KB_NT4_MODES=1,1 KB_NT4_SHAPES=fc1,fc2 timeout 200 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $A
PXA_LIB_PATH=pixart_sigma_amd/variants/lib_nt4_abl16.so KB_NT4_MODES=1,1 KB_NT4_SHAPES=fc1,fc2 timeout 200 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $A
cat $A
