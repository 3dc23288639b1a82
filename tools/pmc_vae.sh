# HBM-traffic counters of the VAE decode kernels (separate --pmc passes, as the microarch guide prescribes).  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r01_pmc_vae.txt
: > $out
for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $ctr | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_vae_$n -o r -- python tools/bench_vae.py --px 512 --batch 8 --iters 1 > /dev/null 2>&1
  { echo "== $ctr : python tools/bench_vae.py --px 512 --batch 8 (SDXL-VAE decode; FETCH_SIZE / WRITE_SIZE in KiB per launch, mean over launches of one grid)"; python tools/pmc_query.py gpurun_out/pmc_vae_$n/r_results.db "gemm_pers|gn_apply|gn_stats|add_kernel"; } >> $out 2>&1
  rm -rf gpurun_out/pmc_vae_$n
done
wc -l $out
