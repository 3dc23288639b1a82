#!/bin/bash
# In-step counters of the token GEMMs: rocprofv3 --pmc over the benchmark's own training step (fp16 build = bench.py's default), one pass per
# counter group, dumped per dispatch.   usage (GPU box, repo root): bash tools/pmc_step.sh TAG
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
tag=${1:-r5}
i=0
# (TCC has 4 slots per pass and FETCH_SIZE takes 3: MI355X_MICROARCH.md "rocprofv3 PMC slots")
for ctr in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 420 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmcs_${tag}_$i -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype --no-configs > gpurun_out/pmcs_${tag}_$i.log 2>&1
  python tools/pmc_step_dump.py gpurun_out/pmcs_${tag}_$i/r_results.db gpurun_out/${tag}_pmc_step_$i.csv "gemm|splitk" >> gpurun_out/pmcs_${tag}_$i.log 2>&1
  tail -2 gpurun_out/pmcs_${tag}_$i.log
  rm -rf gpurun_out/pmcs_${tag}_$i
done
