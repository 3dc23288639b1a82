# GPU-box session: final evidence of round 2 (second session): GPU tier, profile bundle, bench lines (bf16 with both baselines, fp16), inference configs
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $o/r02f_pytest_gpu.log 2>&1
echo "pytest rc $?" >> $o/r02f_pytest_gpu.log
timeout 900 bash tools/profile_round.sh r02f
cp $o/r02f_pmc_attention.json profiles/r02b_pmc_attention.json
timeout 900 python bench.py > $o/r02f_bench_default.json 2> $o/r02f_bench_default.err
timeout 600 python bench.py --dtype fp16 --no-torch-baseline --no-cpu-baseline > $o/r02f_bench_fp16.json 2> $o/r02f_bench_fp16.err
KBENCH_LIBREF=1 timeout 600 python tools/kbench.py all > $o/r02f_kbench.txt 2>&1
timeout 600 python tools/bench_infer.py both > $o/r02f_bench_infer.txt 2>&1
timeout 600 python tools/bench_dmd.py > $o/r02f_bench_dmd.txt 2>&1
tail -3 $o/r02f_pytest_gpu.log; cut -c1-600 $o/r02f_bench_default.json; echo; cut -c1-300 $o/r02f_bench_fp16.json; echo; grep -v amdgpu.ids $o/r02f_kbench.txt | grep -v "split_k=[24]"; grep -v amdgpu.ids $o/r02f_bench_infer.txt | tail -6; grep -v amdgpu.ids $o/r02f_bench_dmd.txt | tail -4
