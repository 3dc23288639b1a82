# GPU-box session r04j: keys-resident cross-attention forward: model suite under the fp16-operand build, inference configs, training step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04j_kvres_model.txt
PXA_OPERAND_DTYPE=f16 timeout 1500 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -2 > $o
timeout 300 python tools/bench_infer.py both 2>&1 | grep workload | cut -c1-330 | sed "s/^/resident : /" >> $o
PXA_ATTN_NO_KVRES=1 timeout 300 python tools/bench_infer.py both 2>&1 | grep workload | cut -c1-330 | sed "s/^/streaming: /" >> $o
for v in resident streaming resident streaming; do
  if [ $v = streaming ]; then export PXA_ATTN_NO_KVRES=1; else unset PXA_ATTN_NO_KVRES; fi
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['dtype'], round(d['ms_per_step'],1), 'ms')" >> $o
done
cat $o
