# GPU-box session r03n: hand-placed dQ kernel: parity (attention tests, all 256 heads at B16), timing A/B, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" > $o/r03n_pytest_attention.log 2>&1
echo "pytest rc $?" >> $o/r03n_pytest_attention.log
timeout 600 python tools/dbg_attn_r03.py grid 2>&1 | grep -v amdgpu.ids > $o/r03n_dbg_grid.txt
for m in 1 0 1 0; do PXA_ATTN_DQ=$m timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids | sed "s/^/dq kernel $m: /"; done > $o/r03n_attn_dq_modes.txt
for m in 1 0; do PXA_ATTN_DQ=$m timeout 300 python tools/kbench.py attn 2>&1 | grep -v amdgpu.ids | grep cross | sed "s/^/dq kernel $m: /"; done >> $o/r03n_attn_dq_modes.txt
for m in 1 0 1 0; do PXA_ATTN_DQ=$m timeout 400 python bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-kernel-roofline --no-other-dtype 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('dq kernel $m: ms_per_step %.1f' % d['ms_per_step'])"; done > $o/r03n_bench_dq_modes.txt
tail -4 $o/r03n_pytest_attention.log; cat $o/r03n_dbg_grid.txt $o/r03n_attn_dq_modes.txt $o/r03n_bench_dq_modes.txt
