# GPU-box session r04c: main-loop schedule probe (probe/dma_bench.hip): 8-wave ping-pong (MODE 3 / 4) vs one wave per SIMD with 128 x 128 per wave (MODE 7), no epilogues
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./probe/dma_bench_bin 8 1 2 > gpurun_out/r04c_sched_probe.txt 2>&1
./probe/dma_bench_bin 8 1 0 >> gpurun_out/r04c_sched_probe.txt 2>&1
cat gpurun_out/r04c_sched_probe.txt
