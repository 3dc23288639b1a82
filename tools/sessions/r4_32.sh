#!/bin/bash
# round 4, session 32: the NT4 / ping-pong / vendor comparison with ROTATING operand sets (no launch finds its own operands or output lines in L2 / MALL - the
# situation inside the training step, where session 31 measured no gain from the NT4 kernel although the repeated-launch bench showed 8-12 %)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
F=$O/r4_32_nt4_rotating.txt
: > $F
for rot in 1 6; do
  KB_NT4_ROTATE=$rot KB_NT4_MODES=0,1,0,1,lib KB_NT4_SHAPES=qkv,proj timeout 300 python tools/kbench_nt4.py time 2>&1 | grep "NT " >> $F
done
cat $F
