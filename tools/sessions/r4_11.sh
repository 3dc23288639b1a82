#!/bin/bash
# round 4, session 11: wider deferred-maximum window (threshold 11, first-tile margin 4): parity on both builds, time vs score range, step profile
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
PXA_OPERAND_DTYPE=f16 timeout 300 python tools/kbench_fwd4.py check > $O/r4_11_fwd4_check_f16.txt 2>&1
PXA_ATTN_FWD4=1 timeout 300 python tools/kbench_fwd4.py check > $O/r4_11_fwd4_check_bf16.txt 2>&1
for qs in 1 2 3; do
  for v in default th6m0 th11m0; do
    if [ $v = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f4_$v.so; fi
    KBENCH_QK_SCALE=$qs PXA_OPERAND_DTYPE=f16 timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=1" | tail -1 >> $O/r4_11_fwd_vs_score_range.txt
  done
done
unset PXA_LIB_PATH
PXA_OPERAND_DTYPE=f16 timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fwd4 or attention" -p no:cacheprovider 2>&1 | tail -3 > $O/r4_11_pytest_attn_f16.txt
timeout 600 python -m pytest tests/test_kernels_gpu.py -q -k "fwd4 or attention" -p no:cacheprovider 2>&1 | tail -3 > $O/r4_11_pytest_attn_bf16.txt
rocprofv3 --kernel-trace --stats -d $O/prof -o step -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $O/r4_11_prof_step.log 2>&1
python tools/export_profile.py $O/prof/step_results.db $O/r4_11_step_kernel_stats.csv 3
rm -rf $O/prof
grep -v amdgpu.ids $O/r4_11_fwd4_check_f16.txt $O/r4_11_fwd4_check_bf16.txt | grep -i "fail\|full grid"; cat $O/r4_11_fwd_vs_score_range.txt $O/r4_11_pytest_attn_f16.txt $O/r4_11_pytest_attn_bf16.txt; head -8 $O/r4_11_step_kernel_stats.csv | cut -c1-130
