# GPU-box session r03c: dK/dV kernels after the explicit vmcnt(0) fix: all-heads grid check, attention tests, per-mode timing, bench A/B, step profile
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 600 python tools/dbg_attn_r03.py grid 2>&1 | grep -v amdgpu.ids > $o/r03c_dbg_grid.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -s -k "attention" > $o/r03c_pytest_attention.log 2>&1
echo "pytest rc $?" >> $o/r03c_pytest_attention.log
for m in 0 1 2; do PXA_ATTN_DKV=$m timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dkv mode $m: /"; done > $o/r03c_attn_dkv_modes.txt
for m in 2 0; do PXA_ATTN_DKV=$m timeout 300 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids | sed "s/^/dkv mode $m: /"; done >> $o/r03c_attn_dkv_modes.txt
PXA_OPERAND_DTYPE=f16 timeout 600 python tools/dbg_attn_r03.py train 0 2 2>&1 | grep -v amdgpu.ids > $o/r03c_dbg_train_f16.txt
for m in 2 0 2 0; do PXA_ATTN_DKV=$m timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-kernel-roofline 2>/dev/null | cut -c1-330 | sed "s/^/dkv mode $m: /"; done > $o/r03c_bench_modes.txt
timeout 900 bash tools/profile_round.sh r03c > /dev/null 2>&1
cat $o/r03c_dbg_grid.txt; tail -3 $o/r03c_pytest_attention.log; cat $o/r03c_attn_dkv_modes.txt; head -8 $o/r03c_dbg_train_f16.txt | cut -c1-200; grep -A8 "mode 2" $o/r03c_dbg_train_f16.txt | cut -c1-200; cut -c1-30,100-330 $o/r03c_bench_modes.txt; head -30 $o/r03c_step_kernel_stats.csv | cut -c1-200
