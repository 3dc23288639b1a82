# GPU-box session r03j: full GPU tier, smoke(), inference benches (text cache on / off), DMD + VAE benches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 3000 python -m pytest tests -m gpu -q > $o/r03j_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $o/r03j_pytest_gpu.log
timeout 600 python __graft_entry__.py smoke > $o/r03j_smoke.log 2>&1
echo "smoke rc $?" >> $o/r03j_smoke.log
timeout 600 python tools/bench_infer.py both > $o/r03j_bench_infer.txt 2>&1
PXA_TEXT_CACHE=0 timeout 600 python tools/bench_infer.py 512 >> $o/r03j_bench_infer.txt 2>&1
timeout 600 python tools/bench_dmd.py > $o/r03j_bench_dmd.txt 2>&1
tail -6 $o/r03j_pytest_gpu.log; tail -4 $o/r03j_smoke.log; grep -v amdgpu.ids $o/r03j_bench_infer.txt | cut -c1-400; grep -v amdgpu.ids $o/r03j_bench_dmd.txt | tail -3
