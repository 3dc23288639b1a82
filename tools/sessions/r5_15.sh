#!/bin/bash
# round 5 session 15: HBM-side counters of the attention kernels INSIDE the training step (roofline.traffic comes from a stand-alone pass: is the step's cache state different?)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
G=$O/r5_15_pmc_step_attention_traffic.txt
echo "$hdr; bench.py --steps 2 --warmup 1 under rocprofv3 --pmc, one pass per counter group; attention kernels of the step, means per launch over the two timed steps (FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them: fetch bytes = 2 x FETCH_SIZE x 1e3 per the gfx950 note)" > $G
for ctr in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  echo "== $ctr" >> $G
  timeout 420 rocprofv3 --kernel-trace --pmc $ctr -d $O/pq15 -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $O/pq15.log 2>&1
  python tools/pmc_step_dump.py $O/pq15/r_results.db $O/pq15.csv "attn_" >> $O/pq15.log 2>&1
  python tools/pmc_step_attn_summary.py $O/pq15.csv >> $G 2>&1
  rm -rf $O/pq15 $O/pq15.csv
done
cat $G | cut -c1-200
