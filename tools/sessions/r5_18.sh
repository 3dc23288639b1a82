#!/bin/bash
# round 5 session 18: items_descending as a feature of the ABI (engine: every token GEMM behind an ascending producer walks downwards): GEMM + model parity on both builds,
# step A/B against PXA_GEMM_ASCENDING=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
for op in f16 bf16; do
  PXA_OPERAND_DTYPE=$op timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" -p no:cacheprovider > $O/r5_18_pytest_gemm_$op.txt 2>&1; echo "rc=$?" >> $O/r5_18_pytest_gemm_$op.txt
done
PXA_OPERAND_DTYPE=f16 timeout 1200 python -m pytest tests/test_model_gpu.py -q -x -k "training or grad or forward_matches" -p no:cacheprovider > $O/r5_18_pytest_model_f16.txt 2>&1; echo "rc=$?" >> $O/r5_18_pytest_model_f16.txt
G=$O/r5_18_step_ab_descending.txt
bash tools/step_ab.sh $G.a "default (token GEMMs behind an ascending producer walk downwards)|A=1" "PXA_GEMM_ASCENDING=1|PXA_GEMM_ASCENDING=1" > /dev/null 2>&1
bash tools/step_ab.sh $G.b "default (token GEMMs behind an ascending producer walk downwards)|A=1" "PXA_GEMM_ASCENDING=1|PXA_GEMM_ASCENDING=1" > /dev/null 2>&1
{ echo "$hdr, bench.py --steps 8 --warmup 3, four rounds"; cat $G.a $G.b; } > $G; rm -f $G.a $G.b
for op in f16 bf16; do tail -2 $O/r5_18_pytest_gemm_$op.txt; done; tail -3 $O/r5_18_pytest_model_f16.txt; cat $G
