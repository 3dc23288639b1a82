# Single implicit-conv layers: time, and HBM-traffic counters (separate --pmc passes), for both K orders.  Run on the GPU box.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/r01_pmc_conv.txt
: > $out
for shape in "128 128 512 512 8" "256 256 256 256 8" "512 512 128 128 8"; do
  for plain in 1 ""; do
    export KBENCH_CONV_PLAIN=$plain
    echo "-- K order: $([ -n "$plain" ] && echo 'segment-major [ky][kx][C]' || echo 'tap-interleaved [C/64][ky][kx][64]')" >> $out
    python tools/kbench_conv.py $shape 20 2>&1 | grep conv3x3 >> $out
    for ctr in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
      n=$(echo $ctr | tr ' ' '_')
      timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d gpurun_out/pmc_c_$n -o r -- python tools/kbench_conv.py $shape 3 > /dev/null 2>&1
      python tools/pmc_query.py gpurun_out/pmc_c_$n/r_results.db "gemm_pers" | grep -E "FETCH|WRITE|TCC" >> $out 2>&1
      rm -rf gpurun_out/pmc_c_$n
    done
  done
done
cat $out
