#!/bin/bash
# round 5 session 16: item order of the NT / NN token GEMMs reversed (each XCD walks its range from the end: the rows a producer wrote last are the ones still in the
# Infinity Cache) - step A/B: none / all / all but the second GEMM of a GEMM -> GEMM pair / only those
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
PXA_OPERAND_DTYPE=f16 PXA_GEMM_REVERSE=1 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm_persistent or gemm_nn_headline or gemm_nt_headline" -p no:cacheprovider > $O/r5_16_pytest.txt 2>&1; echo "rc=$?" >> $O/r5_16_pytest.txt
G=$O/r5_16_step_ab_reverse.txt
bash tools/step_ab.sh $G.tmp "default (ascending)|A=1" "all NT / NN reversed|PXA_GEMM_REVERSE=1" "reversed except K = 4608 (fc2, fc1 dX)|PXA_GEMM_REVERSE=2" "only K = 4608 reversed|PXA_GEMM_REVERSE=3" > /dev/null 2>&1
{ echo "$hdr, bench.py --steps 8 --warmup 3, two rounds"; cat $G.tmp; } > $G; rm -f $G.tmp
tail -3 $O/r5_16_pytest.txt; cat $G
