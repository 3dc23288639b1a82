#!/bin/bash
# round 5 session 17: the reversed item order again (session 16: -1.7 ms for "all NT / NN but the second GEMM of a GEMM -> GEMM pair"): three rounds, and which side carries it
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
G=$O/r5_17_step_ab_reverse.txt
: > $G.all
for rep in 1 2; do
  bash tools/step_ab.sh $G.tmp "default (ascending)|A=1" "reversed except K = 4608|PXA_GEMM_REVERSE=2" "forward (NT) only, except fc2|PXA_GEMM_REVERSE=5" "backward dX (NN) only, except fc1 dX|PXA_GEMM_REVERSE=6" > /dev/null 2>&1
  cat $G.tmp >> $G.all
  [ $rep = 1 ] && sed -i 's/^for rep in 1 2; do/for rep in 1; do/' /dev/null
done
{ echo "$hdr, bench.py --steps 8 --warmup 3, four rounds"; cat $G.all; } > $G; rm -f $G.tmp $G.all
cat $G
