#!/bin/bash
# round 5 session 20: order of a linear layer's two backward GEMMs - dX right behind the kernel that produced dy, then the weight gradient (PXA_DX_FIRST=1) against dW first
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
G=$O/r5_20_step_ab_dx_first.txt
bash tools/step_ab.sh $G.a "default (dW, then dX)|A=1" "dX first (PXA_DX_FIRST=1)|PXA_DX_FIRST=1" > /dev/null 2>&1
bash tools/step_ab.sh $G.b "default (dW, then dX)|A=1" "dX first (PXA_DX_FIRST=1)|PXA_DX_FIRST=1" > /dev/null 2>&1
{ echo "$hdr, bench.py --steps 8 --warmup 3, four rounds"; cat $G.a $G.b; } > $G; rm -f $G.a $G.b
cat $G
