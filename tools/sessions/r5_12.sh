#!/bin/bash
# round 5 session 12: the epilogue ablation ladder of session 1 AGAIN, at the round's HEAD (after GEMM_EPI_EARLY): what the two heavy epilogues still cost, part by part
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1 PXA_OPERAND_DTYPE=f16
V=pixart_sigma_amd/variants
F=$O/r5_12_kbench_epi_ablations.txt
echo "# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16; tools/kbench_epi.py 4 24 (rotating operand sets); GEMM_ABL bits: 8 no column sums, 16 no aux loads, 32 second output not stored, 64 no GELU arithmetic" > $F
for v in "" f16_abl8 f16_abl16 f16_abl32 f16_abl64 f16_abl96 ""; do
  if [ -z "$v" ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$V/lib_$v.so; fi
  timeout 120 python tools/kbench_epi.py 4 24 2>&1 | grep -v amdgpu >> $F
done
cat $F
