# GPU-box session r03g: GEMM without hand-over spills (opaque lane constants) A/B; fp16 delta fold in the dQ kernel; full GPU tier both builds; bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
for lib in default gemm_noopaque default gemm_noopaque; do
  if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=pixart_sigma_amd/variants/lib_$lib.so; fi
  timeout 300 python tools/kbench.py gemm 2>&1 | grep -v amdgpu.ids | grep -v "split_k=[24]"
done > $o/r03g_kbench_gemm_ab.txt
unset PXA_LIB_PATH
timeout 2400 python -m pytest tests -m gpu -x -q > $o/r03g_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $o/r03g_pytest_gpu.log
for lib in default gemm_noopaque default; do
  if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=pixart_sigma_amd/variants/lib_$lib.so; fi
  timeout 400 python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-kernel-roofline --no-other-dtype 2>/dev/null | cut -c1-420 | sed "s|^|$lib: |"
done > $o/r03g_bench_ab.txt
unset PXA_LIB_PATH
timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-kernel-roofline --no-other-dtype 2>/dev/null | cut -c1-700 > $o/r03g_bench_fp16.txt
paste <(grep gemm $o/r03g_kbench_gemm_ab.txt | head -12) <(grep gemm $o/r03g_kbench_gemm_ab.txt | sed -n 13,24p | awk '{print $(NF-1)}') <(grep gemm $o/r03g_kbench_gemm_ab.txt | sed -n 25,36p | awk '{print $(NF-1)}') <(grep gemm $o/r03g_kbench_gemm_ab.txt | sed -n 37,48p | awk '{print $(NF-1)}'); tail -5 $o/r03g_pytest_gpu.log; cut -c1-20,330-420 $o/r03g_bench_ab.txt; cut -c330-700 $o/r03g_bench_fp16.txt
