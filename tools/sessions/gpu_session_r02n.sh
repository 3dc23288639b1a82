# GPU-box session: delta folded into the dP product (bf16 build) and SLP vectorisation of attn.hip: parity + A/B
# Variant libraries: build_variant.py nofold csrc/attn.hip -DATTN_FOLD_DELTA=0; foldnoslp csrc/attn.hip -fno-slp-vectorize; nofoldnoslp csrc/attn.hip -DATTN_FOLD_DELTA=0 -fno-slp-vectorize
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 60 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $o/r02n_pytest_attention.log 2>&1; echo "pytest rc $?" >> $o/r02n_pytest_attention.log
{
  echo "== delta folded (product)"; timeout 60 python tools/kbench_attn_bwd.py
  for v in nofold foldnoslp nofoldnoslp; do echo "== $v"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_$v.so timeout 60 python tools/kbench_attn_bwd.py; done
  echo "== product again"; timeout 60 python tools/kbench_attn_bwd.py
} 2>&1 | grep -v amdgpu.ids > $o/r02n_attn_fold_ab.txt
tail -3 $o/r02n_pytest_attention.log; cat $o/r02n_attn_fold_ab.txt
