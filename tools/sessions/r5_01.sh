#!/bin/bash
# round 5, session 1: (a) this box's baseline step; (b) the two heavy GEMM epilogues taken apart (ablation builds, rotating operands) + two candidate fixes
# (staggered workgroup start, cheaper GELU body); (c) in-step re-decisions of the stand-alone choices VERDICT r04 item 6 lists; (d) in-step counters of the GEMMs (item 3)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1 PXA_OPERAND_DTYPE=f16
V=pixart_sigma_amd/variants
{ echo "# box $(hostname) $(date -u +%FT%TZ) operand f16"; rocm-smi --showproductname 2>/dev/null | grep -i "card series" | head -1; } > $O/r5_01_kbench_epi.txt
for v in "" f16_abl8 f16_abl16 f16_abl32 f16_abl64 f16_abl96 f16_stag9 f16_stag18 f16_gelu2; do
  if [ -z "$v" ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$V/lib_$v.so; fi
  timeout 120 python tools/kbench_epi.py 4 24 2>&1 | grep -v amdgpu >> $O/r5_01_kbench_epi.txt
done
unset PXA_LIB_PATH
unset PXA_OPERAND_DTYPE      # bench.py selects fp16 itself
cfgs=("default|A=1")
for v in stag9 stag18 gelu2 ph16_0 ph16_1 prio0 sto0 opq0 opq2 nt0; do cfgs+=("$v|PXA_LIB_PATH=$V/lib_f16_$v.so"); done
cfgs+=("no_kvres|PXA_ATTN_NO_KVRES=1" "default_again|A=1")
F=$O/r5_01_step_ab.txt
echo "# box $(hostname) $(date -u +%FT%TZ) fp16 build, bench.py --steps 8 --warmup 3, one round" > $F
for cfg in "${cfgs[@]}"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  r=$(env $envs timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline 2>/dev/null \
      | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')
  echo "$label: $r" >> $F
done
timeout 900 bash tools/pmc_step.sh r5_01 > $O/r5_01_pmc_step.log 2>&1
cat $O/r5_01_kbench_epi.txt; cat $F; tail -8 $O/r5_01_pmc_step.log; ls -la $O/r5_01_pmc_step_*.csv
