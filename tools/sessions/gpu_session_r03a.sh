# GPU-box session r03a: corrected MFMA / VALU overlap probe; parity + timing of the three dK/dV kernels; new depth-28 goldens (both operand builds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
( timeout 120 probe/overlap_probe2; timeout 120 probe/overlap_probe2 z ) > $o/r03a_overlap_probe2.txt 2>&1
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "attention" > $o/r03a_pytest_attention.log 2>&1
echo "pytest rc $?" >> $o/r03a_pytest_attention.log
for m in 0 1 2; do PXA_ATTN_DKV=$m timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dkv mode $m: /"; done > $o/r03a_attn_dkv_modes.txt
for m in 0 2; do PXA_ATTN_DKV=$m timeout 300 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids | sed "s/^/dkv mode $m: /"; done >> $o/r03a_attn_dkv_modes.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "full_depth or xl2_1024 or kvevery" > $o/r03a_pytest_model_bf16.log 2>&1
echo "pytest rc $?" >> $o/r03a_pytest_model_bf16.log
PXA_OPERAND_DTYPE=f16 timeout 900 python -m pytest tests/test_model_gpu.py -x -q -s -k "full_depth or xl2_1024 or kvevery" > $o/r03a_pytest_model_f16.log 2>&1
echo "pytest rc $?" >> $o/r03a_pytest_model_f16.log
for m in 0 2; do PXA_ATTN_DKV=$m timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-kernel-roofline 2>/dev/null | cut -c1-400 | sed "s/^/dkv mode $m: /"; done > $o/r03a_bench_modes.txt
head -80 $o/r03a_overlap_probe2.txt; tail -5 $o/r03a_pytest_attention.log; cat $o/r03a_attn_dkv_modes.txt; grep -E "rel-L2|grad err|passed|failed|rc" $o/r03a_pytest_model_bf16.log | head -30; grep -E "rel-L2|grad err|passed|failed|rc" $o/r03a_pytest_model_f16.log | head -30; cat $o/r03a_bench_modes.txt
