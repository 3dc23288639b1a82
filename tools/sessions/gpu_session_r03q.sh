# GPU-box session r03q: reducing backward row kernels (ln_mod_bwd, gate_bwd): LDS-atomic combine -> permlane + plain LDS combine; row map chunked vs interleaved
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "ln_mod or gate_bwd or colsum or final" 2>&1 | tail -3 > $o/r03q_pytest.txt
for rep in 1 2; do
for v in old map0 new; do
  if [ $v = new ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$GRAFT_REPO_ROOT/pixart_sigma_amd/variants/lib_$v.so; fi
  timeout 300 python tools/kbench_elem.py 2>&1 | grep -v amdgpu.ids | grep "bwd" | sed "s/^/$v: /"
done; done > $o/r03q_elem.txt
unset PXA_LIB_PATH
cat $o/r03q_pytest.txt $o/r03q_elem.txt
