# GPU-box session r03l: evidence bundle of the round's code: step profile (both builds), PMC (HBM traffic + SQ), kbench with the vendor yardstick, default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 900 python bench.py > $o/r03l_bench_default.json 2> $o/r03l_bench_default.err
timeout 900 bash tools/profile_round.sh r03l > /dev/null 2>&1
timeout 600 bash tools/pmc_attn.sh > $o/r03l_pmc_attention_sq.txt 2>&1
KBENCH_LIBREF=1 timeout 600 python tools/kbench.py all > $o/r03l_kbench.txt 2>&1
cut -c1-1500 $o/r03l_bench_default.json; echo; head -24 $o/r03l_step_kernel_stats.csv | cut -c1-150; grep -v amdgpu.ids $o/r03l_kbench.txt | grep -v "split_k=[24]"
