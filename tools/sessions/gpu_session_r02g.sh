# GPU-box session: phase structure / priority A/B of the 16-row GEMM main loops
# Variant libraries: build_variant.py ph1 csrc/gemm.hip -DGEMM_PHASE16=1 (one matrix phase per k-unit everywhere); ph0 -DGEMM_PHASE16=0 (two everywhere);
# noprio -DGEMM_M16_PRIO=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
{
  echo "== product (NT two phases, NN / TN one phase per k-unit, s_setprio 1 around the matrix phases)"; timeout 300 python tools/kbench.py gemm
  for v in ph1 ph0 noprio; do echo "== $v"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_$v.so timeout 300 python tools/kbench.py gemm; done
  echo "== product again"; timeout 300 python tools/kbench.py gemm
} > $o/r02g_gemm_phase_prio.txt 2>&1
grep -v amdgpu.ids $o/r02g_gemm_phase_prio.txt | grep -v "split_k=[24]"
