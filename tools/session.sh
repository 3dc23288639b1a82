#!/bin/bash
# ONE parameterised GPU-session runner (round 6; replaces the 112 one-off tools/sessions/*.sh - tools/sessions/MANIFEST.md maps every historical session to the
# recipe + environment that reproduces it).  Runs on the GPU box from the repo root:
#     gpurun --timeout T -- 'bash tools/session.sh TAG recipe [recipe ...]'
# Every text artefact starts with the box, the UTC time, the HEAD the snapshot carries (.gpurun_head, written by tools/gpurun_session.sh) and the operand build.
# A recipe may carry arguments after colons:  step_ab:"default|":"asc|PXA_GEMM_ASCENDING=1"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
tag=$1; shift
export PYTHONUNBUFFERED=1
mkdir -p $O
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown)"
BENCH_QUIET="--no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype --no-configs"

r_tests() {      # the GPU test tier (both operand builds: the tier re-runs the kernel and model suites under f16 in subprocesses)
  echo "$hdr (both operand builds)" > $O/${tag}_pytest_gpu.txt
  timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider "$@" >> $O/${tag}_pytest_gpu.txt 2>&1; echo "pytest rc=$?" >> $O/${tag}_pytest_gpu.txt
  tail -n 5 $O/${tag}_pytest_gpu.txt
}
r_pytest() {     # selected tests:  pytest:tests/test_model_gpu.py:-k:dpm   (PXA_OPERAND_DTYPE from the environment)
  echo "$hdr operand build ${PXA_OPERAND_DTYPE:-bf16}: pytest $*" >> $O/${tag}_pytest_sel.txt
  timeout ${PYTEST_TIMEOUT:-600} python -m pytest -m gpu -q -p no:cacheprovider -s "$@" >> $O/${tag}_pytest_sel.txt 2>&1; echo "pytest rc=$?" >> $O/${tag}_pytest_sel.txt
  grep -v amdgpu $O/${tag}_pytest_sel.txt | tail -n 40 | cut -c1-300
}
r_pytest_f16() { PXA_OPERAND_DTYPE=f16 r_pytest "$@"; }     # the same under the fp16-operand library
r_smoke() {
  echo "$hdr" > $O/${tag}_smoke.txt
  timeout 900 python __graft_entry__.py smoke >> $O/${tag}_smoke.txt 2>&1; echo "smoke rc=$?" >> $O/${tag}_smoke.txt
  grep -v amdgpu $O/${tag}_smoke.txt | tail -n 12 | cut -c1-600
}
r_bench() {      # the default bench line with every leg (what the driver runs)
  timeout 2400 python bench.py "$@" > $O/${tag}_bench_default.json 2> $O/${tag}_bench_default.err
  cut -c1-1500 $O/${tag}_bench_default.json; tail -n 3 $O/${tag}_bench_default.err
}
r_profile_step() {   # rocprofv3 --kernel-trace --stats of the benchmark's training step -> per-kernel csv + family table
  rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o step -- python bench.py --steps 2 --warmup 1 $BENCH_QUIET > $O/prof_${tag}_step.log 2>&1
  python tools/export_profile.py $O/prof_$tag/step_results.db $O/${tag}_step_kernel_stats.csv 3
  rm -rf $O/prof_$tag
  { echo "$hdr operand build f16"; python tools/family_times.py $O/${tag}_step_kernel_stats.csv; } > $O/${tag}_family_times.txt
  cat $O/${tag}_family_times.txt; head -14 $O/${tag}_step_kernel_stats.csv | cut -c1-130
}
r_profile_round() { timeout 900 bash tools/profile_round.sh $tag > $O/${tag}_profile_round.log 2>&1; tail -n 5 $O/${tag}_profile_round.log; }
r_pmc_step() {
  timeout 1200 bash tools/pmc_step.sh $tag > $O/${tag}_pmc_step.log 2>&1
  { echo "$hdr operand build f16"; python tools/pmc_step_table.py $O/$tag; } > $O/${tag}_pmc_step_gemm_table.txt 2>&1
  cut -c1-150 $O/${tag}_pmc_step_gemm_table.txt
}
r_step_ab() {    # same-box A/B of the training step:  step_ab:"label|ENV=1 ENV2=x":"other|"
  echo "$hdr operand build f16" > $O/${tag}_step_ab.hdr
  bash tools/step_ab.sh $O/${tag}_step_ab.body "$@" > /dev/null
  cat $O/${tag}_step_ab.hdr $O/${tag}_step_ab.body > $O/${tag}_step_ab.txt; rm -f $O/${tag}_step_ab.hdr $O/${tag}_step_ab.body
  cat $O/${tag}_step_ab.txt
}
r_infer() {      # BASELINE configs 2 / 4 / 5, fp16 operands (what bench.py's `configs` leg runs)
  echo "$hdr operand build f16 (tools/bench_infer.py, tools/bench_dmd.py) $*" >> $O/${tag}_bench_infer.txt
  timeout 900 python tools/bench_infer.py both "$@" >> $O/${tag}_bench_infer.txt 2>&1
  timeout 900 python tools/bench_dmd.py >> $O/${tag}_bench_infer.txt 2>&1
  grep -v amdgpu $O/${tag}_bench_infer.txt | tail -n 4 | cut -c1-500
}
r_profile_infer() {  # kernel trace of the denoiser evaluations of config 2 (512px, model batch 16) and config 4 (2K, model batch 4), per NFE
  export PXA_OPERAND_DTYPE=f16
  for c in "512|4" "2k|2"; do
    w=${c%%|*}; st=${c#*|}
    rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_i$w -o r -- python tools/bench_infer.py $w --steps $st > $O/prof_${tag}_i$w.log 2>&1
    python tools/export_profile.py $O/prof_${tag}_i$w/r_results.db $O/${tag}_infer${w}_kernel_stats.csv $((3 * st))     # three sample() calls of `st` evaluations
    rm -rf $O/prof_${tag}_i$w
    sed -i "1i $hdr operand build f16; python tools/bench_infer.py $w --steps $st; per denoiser evaluation (NFE)" $O/${tag}_infer${w}_kernel_stats.csv
    head -16 $O/${tag}_infer${w}_kernel_stats.csv | cut -c1-140
  done
}
r_profile_vae() {    # kernel trace of the SD-VAE decode of config 5 (64 x 512px): per-kernel csv + the launch-ordered trace of the last decode (per-layer table)
  export PXA_OPERAND_DTYPE=f16
  rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_vae -o r -- python tools/bench_vae.py --px 512 --batch ${1:-64} --iters 2 > $O/prof_${tag}_vae.log 2>&1
  python tools/export_profile.py $O/prof_${tag}_vae/r_results.db $O/${tag}_vae_decode512_kernel_stats.csv 3
  python tools/export_trace.py $O/prof_${tag}_vae/r_results.db $O/${tag}_vae_decode512_trace.csv 3
  rm -rf $O/prof_${tag}_vae
  sed -i "1i $hdr operand build f16; python tools/bench_vae.py --px 512 --batch ${1:-64} --iters 2; per decode" $O/${tag}_vae_decode512_kernel_stats.csv
  grep -v amdgpu $O/prof_${tag}_vae.log | tail -n 2 | cut -c1-400; head -14 $O/${tag}_vae_decode512_kernel_stats.csv | cut -c1-140
  python tools/vae_layer_table.py $O/${tag}_vae_decode512_trace.csv ${1:-64} > $O/${tag}_vae_layer_table.txt 2>&1 && sed -i "1i $hdr operand build f16" $O/${tag}_vae_layer_table.txt
  cat $O/${tag}_vae_layer_table.txt | cut -c1-170
}
r_dp_ab() {      # the N > 1 code path as far as one GPU can run it: bench.py under torchrun (world 1, real RCCL group) with PXA_DP_FORCE_COLLECTIVES=1 - every gradient
                 # bucket all-reduced from the engine's hooks - against the plain run; dynamic item cursors (the reducer's default) against the static split
                 # (PXA_DP_STATIC_ITEMS=1); PXA_DP_TRACE=1 prints the per-bucket record.  Two alternating rounds, 8 steps after 3 of warm-up.
  echo "$hdr operand build f16" > $O/${tag}_dp_ab.txt
  port=29650
  for rep in 1 2; do
    for cfg in "plain (no process group)|NOPG=1" "rccl world 1, forced collectives, dynamic cursors|PXA_DP_FORCE_COLLECTIVES=1 PXA_DP_TRACE=1" "rccl world 1, forced collectives, static split|PXA_DP_FORCE_COLLECTIVES=1 PXA_DP_TRACE=1 PXA_DP_STATIC_ITEMS=1"; do
      label=${cfg%%|*}; envs=${cfg#*|}; port=$((port + 1))
      if [ "$envs" = "NOPG=1" ]; then launcher="python"; else launcher="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $port"; fi
      r=$(env $envs timeout 600 $launcher bench.py --gpus 1 --steps 8 --warmup 3 $BENCH_QUIET 2> $O/${tag}_dp_ab.err | python -c 'import sys,json
d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); t=d.get("dp_trace") or {}
print(round(d["ms_per_step"],2), "ms/step; process_group", d["process_group"], "; trace:", {k: t[k] for k in ("backward_done_ms","all_reduced_ms","exposed_ms","world") if k in t}, [(b["bucket"], b["launched_from"]) for b in t.get("buckets", [])][-2:])')
      echo "$label: $r" >> $O/${tag}_dp_ab.txt
      grep "dp trace" $O/${tag}_dp_ab.err | tail -1 >> $O/${tag}_dp_ab.txt
    done
  done
  cat $O/${tag}_dp_ab.txt
}
r_pmc_vae() {    # fabric-traffic and matrix-pipe counters of the VAE decode kernels (separate --pmc passes, as the microarch guide prescribes), fp16 build
  export PXA_OPERAND_DTYPE=f16
  b=${1:-16}
  echo "$hdr operand build f16; python tools/bench_vae.py --px 512 --batch $b --iters 1 under rocprofv3 --pmc; FETCH_SIZE / WRITE_SIZE in KB per launch (FETCH_SIZE x 2 on gfx950), means over the launches of one (kernel, grid)" > $O/${tag}_pmc_vae.txt
  for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
    n=$(echo $ctr | tr ' ' '_')
    timeout 400 rocprofv3 --kernel-trace --pmc $ctr -d $O/pmc_${tag}_v$n -o r -- python tools/bench_vae.py --px 512 --batch $b --iters 1 > /dev/null 2>&1
    { echo "== $ctr"; python tools/pmc_query.py $O/pmc_${tag}_v$n/r_results.db "gemm_pers|gn_apply|conv3x3_small|softmax_rows|gemm_glds"; } >> $O/${tag}_pmc_vae.txt 2>&1
    rm -rf $O/pmc_${tag}_v$n
  done
  head -60 $O/${tag}_pmc_vae.txt | cut -c1-200
}
r_run() {        # anything else, logged under the tag:  run:python:tools/kbench.py:attn
  echo "$hdr operand build ${PXA_OPERAND_DTYPE:-bf16}: $*" >> $O/${tag}_run.txt
  timeout ${RUN_TIMEOUT:-600} "$@" >> $O/${tag}_run.txt 2>&1; echo "rc=$?" >> $O/${tag}_run.txt
  grep -v amdgpu $O/${tag}_run.txt | tail -n 4 | cut -c1-400
}

for spec in "$@"; do
  IFS=':' read -r -a parts <<< "$spec"
  name=${parts[0]}
  echo "=== $name ${parts[*]:1}"
  "r_$name" "${parts[@]:1}"
done
