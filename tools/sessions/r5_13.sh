#!/bin/bash
# round 5 session 13: GEMM_AUX_TOUCH (the aux flavours request their aux sub-tile into L2 from inside the main loop): GEMM parity on both builds, the heavy-epilogue
# bench against the no-touch build and other touch distances, step A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
V=pixart_sigma_amd/variants
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
for op in f16 bf16; do
  PXA_OPERAND_DTYPE=$op timeout 900 python -m pytest tests/test_kernels_gpu.py -q -x -k "gemm" -p no:cacheprovider > $O/r5_13_pytest_gemm_$op.txt 2>&1; echo "rc=$?" >> $O/r5_13_pytest_gemm_$op.txt
done
F=$O/r5_13_kbench_epi_touch.txt
echo "$hdr; tools/kbench_epi.py 4 24 (rotating operand sets); default = GEMM_AUX_TOUCH 8" > $F
export PXA_OPERAND_DTYPE=f16
for v in "" f16_touch0 f16_touch6 f16_touch12 f16_touch16 "" f16_touch0; do
  if [ -z "$v" ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$V/lib_$v.so; fi
  timeout 120 python tools/kbench_epi.py 4 24 2>&1 | grep -v amdgpu | grep -E "lib:|fc2 dX" >> $F
done
unset PXA_LIB_PATH PXA_OPERAND_DTYPE
G=$O/r5_13_step_ab_touch.txt
bash tools/step_ab.sh $G.tmp "default (GEMM_AUX_TOUCH 8)|A=1" "no touch|PXA_LIB_PATH=$V/lib_f16_touch0.so" > /dev/null 2>&1
{ echo "$hdr, bench.py --steps 8 --warmup 3, two rounds"; cat $G.tmp; } > $G; rm -f $G.tmp
for op in f16 bf16; do tail -3 $O/r5_13_pytest_gemm_$op.txt; done; cat $F; cat $G
