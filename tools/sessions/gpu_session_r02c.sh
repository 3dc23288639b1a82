# GPU-box session: v_permlane16_swap probe, parity of the 16x16x32 second products of the attention kernels, their A/B against the 32-row
# build (-DATTN_PV16=0), the GEMM main loop with its FLOPs issued as 16x16x32 MFMAs (-DGEMM_ABL=4, wrong results on purpose), full GPU tier.
# Variant libraries of this session (state of commit d2156f7's parent): build_variant.py pv32 csrc/attn.hip -DATTN_PV16=0 (then: every attention kernel);
# build_variant.py gemm16 csrc/gemm.hip -DGEMM_ABL=4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 60 ./probe/perm16 > $o/r02c_perm16.txt 2>&1
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $o/r02c_pytest_attention.log 2>&1
echo "pytest rc $?" >> $o/r02c_pytest_attention.log
{
  echo "== 16x16x32 second products (product)"; timeout 300 python tools/kbench_attn_bwd.py
  echo "== 32-row tiles (-DATTN_PV16=0)"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_pv32.so timeout 300 python tools/kbench_attn_bwd.py
  echo "== product, all attention shapes"; timeout 300 python tools/kbench.py attn
  echo "== 32-row tiles, all attention shapes"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_pv32.so timeout 300 python tools/kbench.py attn
  echo "== product again"; timeout 300 python tools/kbench_attn_bwd.py
} > $o/r02c_attn_pv16_ab.txt 2>&1
{
  echo "== GEMMs, product (32x32x16)"; timeout 300 python tools/kbench.py gemm
  echo "== GEMMs, main-loop FLOPs as 16x16x32 MFMAs (ablation, wrong results)"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_gemm16.so timeout 300 python tools/kbench.py gemm
} > $o/r02c_gemm_mfma16_ablation.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $o/r02c_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $o/r02c_pytest_gpu.log
timeout 600 python bench.py --no-torch-baseline --no-cpu-baseline > $o/r02c_bench_default.json 2> $o/r02c_bench_default.err
cat $o/r02c_perm16.txt; tail -5 $o/r02c_pytest_attention.log; grep -v amdgpu.ids $o/r02c_attn_pv16_ab.txt; grep -v amdgpu.ids $o/r02c_gemm_mfma16_ablation.txt | grep -v "split_k=[24]"; tail -4 $o/r02c_pytest_gpu.log; cut -c1-300 $o/r02c_bench_default.json
