# GPU-box session r03b: which dK/dV kernel is off on the full B16 grid; depth-28 training golden per mode (three-term lse / delta split); bench order reversed
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 600 python tools/dbg_attn_r03.py grid 2>&1 | grep -v amdgpu.ids > $o/r03b_dbg_grid.txt
timeout 600 python tools/dbg_attn_r03.py train 0 1 2 2>&1 | grep -v amdgpu.ids > $o/r03b_dbg_train_bf16.txt
PXA_OPERAND_DTYPE=f16 timeout 600 python tools/dbg_attn_r03.py train 0 2 2>&1 | grep -v amdgpu.ids > $o/r03b_dbg_train_f16.txt
for m in 2 0 2 0; do PXA_ATTN_DKV=$m timeout 400 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-torch-baseline --no-kernel-roofline 2>/dev/null | cut -c1-330 | sed "s/^/dkv mode $m: /"; done > $o/r03b_bench_modes.txt
cat $o/r03b_dbg_grid.txt $o/r03b_dbg_train_bf16.txt $o/r03b_dbg_train_f16.txt; cut -c1-30,100-330 $o/r03b_bench_modes.txt
