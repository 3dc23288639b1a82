# GPU-box session: same-box A/B of the non-temporal accesses per source (optimizer, delta pre-pass, row kernels)
# Variant libraries: build_variant.py {optnt0,attnnt0,normnt0} csrc/{optim,attn,norm}.hip -DPXA_STREAM_NT=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
{
  echo "== optimizer, product (non-temporal)"; timeout 200 python tools/bench_opt.py
  echo "== optimizer, plain accesses"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_optnt0.so timeout 200 python tools/bench_opt.py
  echo "== optimizer, product again"; timeout 200 python tools/bench_opt.py
  echo "== attention backward, product (delta pre-pass with non-temporal loads)"; timeout 200 python tools/kbench_attn_bwd.py
  echo "== attention backward, plain loads"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_attnnt0.so timeout 200 python tools/kbench_attn_bwd.py
  echo "== row kernels, product"; timeout 200 python tools/kbench_elem.py
  echo "== row kernels, plain accesses"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_normnt0.so timeout 200 python tools/kbench_elem.py
  echo "== row kernels, product again"; timeout 200 python tools/kbench_elem.py
} 2>&1 | grep -v amdgpu.ids > $o/r02j_nt_ab.txt
cat $o/r02j_nt_ab.txt
