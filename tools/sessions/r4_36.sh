#!/bin/bash
# round 4, session 36: the round profile at the final HEAD (dK/dV on attn_bwd_dkv4_kernel): per-kernel statistics of the benchmark step and the FETCH_SIZE / WRITE_SIZE
# passes of the attention kernels (the TCC and GEMM passes of tools/profile_round.sh are unchanged since the bundle: profiles/r4final_pmc_gemm.txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=r4final
timeout 150 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o step -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > gpurun_out/prof_${tag}_step.log 2>&1
python tools/export_profile.py gpurun_out/prof_$tag/step_results.db gpurun_out/${tag}_step_kernel_stats.csv 3
rm -rf gpurun_out/prof_$tag
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 100 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_${tag}_$ctr -o r -- python tools/kbench.py attn > /dev/null 2>&1
  { echo "== $ctr : python tools/kbench.py attn (self-attention B16 H16 N4096 d72; cross-attention L300)"; python tools/pmc_query.py gpurun_out/pmc_${tag}_$ctr/r_results.db "attn" --json gpurun_out/${tag}_pmc_attention.json; } >> gpurun_out/${tag}_pmc_attention.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_$ctr
done
head -6 gpurun_out/${tag}_step_kernel_stats.csv | cut -c1-110; grep -c dkv4 gpurun_out/${tag}_pmc_attention.json
