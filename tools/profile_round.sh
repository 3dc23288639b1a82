# Round profile bundle (run on the GPU box from the repo root): per-kernel stats of the benchmark step + HBM-traffic counters of the
# dominant kernels.  Writes text/CSV summaries under gpurun_out/; copy the ones to keep into profiles/ (tools/collect_bundle.sh).
# Everything here runs the fp16-operand build - the one bench.py times (the r4 PMC passes had run the bf16 library).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r02}
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$tag -o step -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype --no-configs > gpurun_out/prof_${tag}_step.log 2>&1
python tools/export_profile.py gpurun_out/prof_$tag/step_results.db gpurun_out/${tag}_step_kernel_stats.csv 3
rm -rf gpurun_out/prof_$tag
export PXA_OPERAND_DTYPE=f16
echo "$hdr" > gpurun_out/${tag}_pmc_attention.txt
echo "$hdr" > gpurun_out/${tag}_pmc_gemm.txt
rm -f gpurun_out/${tag}_pmc_attention.json
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $ctr | tr ' ' '_')
  rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_${tag}_$n -o r -- python tools/kbench.py attn > /dev/null 2>&1
  { echo "== $ctr : python tools/kbench.py attn (self-attention B16 H16 N4096 d72, q prescaled; cross-attention L300)"; python tools/pmc_query.py gpurun_out/pmc_${tag}_$n/r_results.db "attn" --json gpurun_out/${tag}_pmc_attention.json; } >> gpurun_out/${tag}_pmc_attention.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_$n
  rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_${tag}_g$n -o r -- python tools/kbench_one.py 65536 4608 1152 NT 5 > /dev/null 2>&1
  { echo "== $ctr : python tools/kbench_one.py 65536 4608 1152 NT (fc1 forward shape, 16-bit output)"; python tools/pmc_query.py gpurun_out/pmc_${tag}_g$n/r_results.db "gemm"; } >> gpurun_out/${tag}_pmc_gemm.txt 2>&1
  rm -rf gpurun_out/pmc_${tag}_g$n
done
