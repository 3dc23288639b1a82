# GPU-box session: streaming (non-temporal) accesses as the default of the row / optimizer / delta kernels: kernel parity, optimizer time, step A/B of
# non-temporal GEMM output stores.  Variant library: build_variant.py gemmnt csrc/gemm.hip -DGEMM_NT_STORE=1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_came_gpu.py -q > $o/r02i_pytest_kernels.log 2>&1
echo "pytest rc $?" >> $o/r02i_pytest_kernels.log
timeout 200 python tools/bench_opt.py > $o/r02i_bench_opt.txt 2>&1
timeout 200 python tools/kbench_elem.py > $o/r02i_elem.txt 2>&1
{
  echo "== bench product"; timeout 300 python bench.py --no-torch-baseline --no-cpu-baseline
  echo "== bench gemmnt"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_gemmnt.so timeout 300 python bench.py --no-torch-baseline --no-cpu-baseline --no-kernel-roofline
  echo "== bench product again"; timeout 300 python bench.py --no-torch-baseline --no-cpu-baseline --no-kernel-roofline
} 2>&1 | grep -v amdgpu.ids > $o/r02i_bench.txt
tail -4 $o/r02i_pytest_kernels.log; grep -v amdgpu.ids $o/r02i_bench_opt.txt | tail -6; grep -v amdgpu.ids $o/r02i_elem.txt; cut -c1-330 $o/r02i_bench.txt
