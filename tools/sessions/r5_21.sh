#!/bin/bash
# round 5 session 21: kernel trace of the training step at the round's last code commit (token GEMMs walking downwards) + family times
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out; tag=r5head2
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
for cfg in "desc|A=1" "asc|PXA_GEMM_ASCENDING=1"; do
  label=${cfg%%|*}; envs=${cfg#*|}
  env $envs rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_$label -o step -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $O/prof_${tag}_$label.log 2>&1
  python tools/export_profile.py $O/prof_${tag}_$label/step_results.db $O/${tag}_step_kernel_stats_$label.csv 3
  rm -rf $O/prof_${tag}_$label
done
{ echo "$hdr; default (token GEMMs behind an ascending producer walk downwards), then PXA_GEMM_ASCENDING=1"; for l in desc asc; do python tools/family_times.py $O/${tag}_step_kernel_stats_$l.csv; done; } > $O/${tag}_family_times.txt
cat $O/${tag}_family_times.txt; for l in desc asc; do grep -E "gemm_pers_kernel<(0|1)," $O/${tag}_step_kernel_stats_$l.csv | cut -c1-100; done
