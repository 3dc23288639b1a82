# GPU-box session: NT GEMMs on 16x16x32 MFMAs (parity + A/B against -DGEMM_NT16=0), dK/dV kernel modes (-DATTN_DKV16=1..3), full GPU tier, bench.
# Variant libraries of this session: build_variant.py nt32 csrc/gemm.hip -DGEMM_NT16=0 (the switch is GEMM_M16 since the next commit); build_variant.py dkv{1,2,3}
# csrc/attn.hip -DATTN_DKV16={1,2,3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "gemm" > $o/r02d_pytest_gemm.log 2>&1
echo "pytest rc $?" >> $o/r02d_pytest_gemm.log
{
  echo "== NT on 16x16x32 (product)"; timeout 300 python tools/kbench.py gemm
  echo "== NT on 32x32x16 (-DGEMM_NT16=0)"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_nt32.so timeout 300 python tools/kbench.py gemm
} > $o/r02d_gemm_nt16_ab.txt 2>&1
{
  echo "== dK/dV mode 0 (product: 32-row tiles)"; timeout 300 python tools/kbench_attn_bwd.py
  for m in 1 2 3; do echo "== dK/dV mode $m"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_dkv$m.so timeout 300 python tools/kbench_attn_bwd.py; done
  echo "== product again"; timeout 300 python tools/kbench_attn_bwd.py
} > $o/r02d_attn_dkv_modes.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $o/r02d_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $o/r02d_pytest_gpu.log
timeout 600 python bench.py --no-torch-baseline --no-cpu-baseline > $o/r02d_bench_default.json 2> $o/r02d_bench_default.err
tail -5 $o/r02d_pytest_gemm.log; grep -v amdgpu.ids $o/r02d_gemm_nt16_ab.txt | grep -v "split_k=[24]"; grep -v amdgpu.ids $o/r02d_attn_dkv_modes.txt; tail -4 $o/r02d_pytest_gpu.log; cut -c1-300 $o/r02d_bench_default.json
