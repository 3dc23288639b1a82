#!/bin/bash
# round 5 session 10: (a) the one in-step re-decision of session 1 that was outside the noise in a single run (workgroups on odd CU slots start 90 ns late, GEMM_STAGGER=9:
# -2.2 ms) repeated with three alternating rounds; (b) the benchmark under its multi-process launcher at world size 1 (python -m torch.distributed.run, RCCL group of one)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
export PYTHONUNBUFFERED=1
hdr="# box $(hostname) $(date -u +%FT%TZ) HEAD $(cat .gpurun_head 2>/dev/null || echo unknown) operand build f16"
F=$O/r5_10_step_ab_stagger.txt
bash tools/step_ab.sh $F.a "default|A=1" "GEMM_STAGGER=9|PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f16_stag9.so" > /dev/null 2>&1
bash tools/step_ab.sh $F.b "default|A=1" "GEMM_STAGGER=9|PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f16_stag9.so" > /dev/null 2>&1
{ echo "$hdr, bench.py --steps 8 --warmup 3, four alternating rounds"; cat $F.a $F.b; } > $F; rm -f $F.a $F.b
G=$O/r5_10_bench_torchrun_world1.txt
echo "$hdr; python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 (no baselines)" > $G
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype 2>/dev/null | tail -1 | cut -c1-1200 >> $G
cat $F; cat $G | cut -c1-700
