# GPU-box session: NN / TN GEMMs and the VAE's implicit convolutions on 16x16x32 MFMAs (GEMM_M16=7) against NT only (-DGEMM_M16=1): parity, A/B, VAE
# Variant libraries of this session (product built with GEMM_M16=7): build_variant.py m16nt csrc/gemm.hip -DGEMM_M16=1; build_variant.py m16none csrc/gemm.hip -DGEMM_M16=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_vae_gpu.py -q -k "gemm or vae or conv" > $o/r02e_pytest_gemm_vae.log 2>&1
echo "pytest rc $?" >> $o/r02e_pytest_gemm_vae.log
{
  echo "== all layouts on 16x16x32 (product)"; timeout 300 python tools/kbench.py gemm
  echo "== NT only (-DGEMM_M16=1)"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_m16nt.so timeout 300 python tools/kbench.py gemm
  echo "== product again"; timeout 300 python tools/kbench.py gemm
} > $o/r02e_gemm_m16_ab.txt 2>&1
{
  echo "== VAE decode 512px x 16, product"; timeout 300 python tools/bench_vae.py --px 512 --batch 16
  echo "== VAE decode 512px x 16, every layout on 32x32x16 (-DGEMM_M16=0)"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_m16none.so timeout 300 python tools/bench_vae.py --px 512 --batch 16
} > $o/r02e_vae.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $o/r02e_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $o/r02e_pytest_gpu.log
timeout 600 python bench.py --no-torch-baseline --no-cpu-baseline > $o/r02e_bench_default.json 2> $o/r02e_bench_default.err
tail -5 $o/r02e_pytest_gemm_vae.log; grep -v amdgpu.ids $o/r02e_gemm_m16_ab.txt | grep -v "split_k=[24]"; grep -v amdgpu.ids $o/r02e_vae.txt | tail -12; tail -4 $o/r02e_pytest_gpu.log; cut -c1-300 $o/r02e_bench_default.json
