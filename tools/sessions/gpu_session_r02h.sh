# GPU-box session: non-temporal hints on the row kernels' streams (norm.hip -DELEM_NT=1 stores / 2 loads / 3 both), one-sub-tile forward for cross-attention
# Variant libraries: build_variant.py nt{1,2,3} csrc/norm.hip -DELEM_NT={1,2,3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
{
  echo "== product"; timeout 200 python tools/kbench_elem.py
  for v in nt1 nt2 nt3; do echo "== $v"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_$v.so timeout 200 python tools/kbench_elem.py; done
  echo "== product again"; timeout 200 python tools/kbench_elem.py
} > $o/r02h_elem_nt.txt 2>&1
{
  echo "== product"; timeout 200 python tools/kbench.py attn
  echo "== PXA_ATTN_FWD1=1 (one query sub-tile per wave)"; PXA_ATTN_FWD1=1 timeout 200 python tools/kbench.py attn
} > $o/r02h_attn_fwd1.txt 2>&1
{
  echo "== bench product"; timeout 300 python bench.py --no-torch-baseline --no-cpu-baseline --no-kernel-roofline
  for v in nt1 nt3; do echo "== bench $v"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_$v.so timeout 300 python bench.py --no-torch-baseline --no-cpu-baseline --no-kernel-roofline; done
  echo "== bench product again"; timeout 300 python bench.py --no-torch-baseline --no-cpu-baseline --no-kernel-roofline
} 2>&1 | grep -v amdgpu.ids | cut -c1-330 > $o/r02h_bench_nt.txt
grep -v amdgpu.ids $o/r02h_elem_nt.txt; grep -v amdgpu.ids $o/r02h_attn_fwd1.txt; cat $o/r02h_bench_nt.txt
