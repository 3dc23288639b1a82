# GPU-box session r03e: LDS-DMA as inline asm (compiler no longer forces vmcnt(0) before transpose reads): parity of the whole kernel + model tier,
# same-box A/B of GEMMs and attention against the builtin-DMA variants, bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_vae_gpu.py -x -q > $o/r03e_pytest.log 2>&1
echo "pytest rc $?" >> $o/r03e_pytest.log
for lib in default pixart_sigma_amd/variants/lib_gemm_builtin_dma.so default pixart_sigma_amd/variants/lib_gemm_builtin_dma.so; do
  if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$lib; fi
  timeout 300 python tools/kbench.py gemm 2>&1 | grep -v amdgpu.ids | grep -v "split_k=[24]"
done > $o/r03e_kbench_gemm_ab.txt
for lib in default pixart_sigma_amd/variants/lib_attn_builtin_dma.so default; do
  if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$lib; fi
  timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids
done > $o/r03e_kbench_attn_ab.txt
unset PXA_LIB_PATH
for lib in default pixart_sigma_amd/variants/lib_gemm_builtin_dma.so default; do
  if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$lib; fi
  timeout 400 python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-kernel-roofline --no-other-dtype 2>/dev/null | cut -c1-330 | sed "s|^|$lib: |"
done > $o/r03e_bench_ab.txt
unset PXA_LIB_PATH
tail -4 $o/r03e_pytest.log; cat $o/r03e_kbench_gemm_ab.txt; cat $o/r03e_kbench_attn_ab.txt; cut -c1-60,130-330 $o/r03e_bench_ab.txt
