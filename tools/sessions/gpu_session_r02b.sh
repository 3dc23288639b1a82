# One GPU-box session (run from the repo root through gpurun): MFMA shape probe, A/B of the XCD-aware attention block order, the GPU test
# tier, the default bench line and the round profile bundle.  Everything lands under gpurun_out/; copy what is kept into profiles/.
# Variant library of this session: python tools/build_variant.py noxcd pixart_sigma_amd/csrc/attn.hip -DATTN_XCD_HEADS=0
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
{ echo "== probe/mfma_power (N(0,1) operands)"; timeout 60 ./probe/mfma_power 150; echo "== zeros"; timeout 60 ./probe/mfma_power 150 1; } > $o/r02b_mfma_power.txt 2>&1
{
  echo "== XCD-aware block order (product)"; timeout 300 python tools/kbench_attn_bwd.py
  echo "== plain block order (-DATTN_XCD_HEADS=0)"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_noxcd.so timeout 300 python tools/kbench_attn_bwd.py
  echo "== product, all attention shapes"; timeout 300 python tools/kbench.py attn
  echo "== plain order, all attention shapes"; PXA_LIB_PATH=pixart_sigma_amd/variants/lib_noxcd.so timeout 300 python tools/kbench.py attn
  echo "== product again (box drift check)"; timeout 300 python tools/kbench_attn_bwd.py
} > $o/r02b_attn_xcd_ab.txt 2>&1
timeout 900 python -m pytest tests -m gpu -x -q > $o/r02b_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $o/r02b_pytest_gpu.log
timeout 900 bash tools/profile_round.sh r02b
cp $o/r02b_pmc_attention.json profiles/r02b_pmc_attention.json      # bench.py reads roofline.traffic from the newest committed PMC file
timeout 600 python bench.py --no-torch-baseline --no-cpu-baseline > $o/r02b_bench_default.json 2> $o/r02b_bench_default.err
tail -3 $o/r02b_pytest_gpu.log; cat $o/r02b_mfma_power.txt; cat $o/r02b_attn_xcd_ab.txt | grep -v amdgpu.ids; cut -c1-400 $o/r02b_bench_default.json
