# GPU-box session r03f: dK/dV kernel variants (priority, one wave per SIMD, slot order), SQ counters of the attention kernels, the new default bench
# line end to end (fp16 + other_dtype + baselines), secondary benches (grad checkpointing, ragged captions)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
for lib in default dkv2_prio1 dkv2_prio3 dkv2_w1 dkv2_nofirst default; do
  if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=pixart_sigma_amd/variants/lib_$lib.so; fi
  timeout 300 python tools/kbench_attn_bwd.py 2>&1 | grep -v amdgpu.ids
done > $o/r03f_dkv2_variants.txt
unset PXA_LIB_PATH
timeout 600 bash tools/pmc_attn.sh > $o/r03f_pmc_attention_sq.txt 2>&1
timeout 900 python bench.py > $o/r03f_bench_default.json 2> $o/r03f_bench_default.err
timeout 400 python bench.py --dtype bf16 --grad-checkpoint --no-cpu-baseline --no-torch-baseline --no-kernel-roofline --no-other-dtype > $o/r03f_bench_gradckpt_bf16.json 2>/dev/null
timeout 400 python bench.py --dtype bf16 --ragged-text --no-cpu-baseline --no-torch-baseline --no-kernel-roofline --no-other-dtype > $o/r03f_bench_ragged_bf16.json 2>/dev/null
timeout 400 python bench.py --dtype fp16 --ragged-text --no-cpu-baseline --no-torch-baseline --no-kernel-roofline --no-other-dtype > $o/r03f_bench_ragged_fp16.json 2>/dev/null
cat $o/r03f_dkv2_variants.txt; grep -A9 "dkv2" $o/r03f_pmc_attention_sq.txt | head -60; cat $o/r03f_bench_default.json; tail -3 $o/r03f_bench_default.err; cut -c1-420 $o/r03f_bench_gradckpt_bf16.json; echo; cut -c1-420 $o/r03f_bench_ragged_bf16.json; echo; cut -c1-420 $o/r03f_bench_ragged_fp16.json
