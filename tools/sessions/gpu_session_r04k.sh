# GPU-box session r04k: keys-resident dQ kernel with the delta pre-pass folded in: parity (kernels, model suite in both builds), cross-attention time, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04k_kvres_dq.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -3 > $o
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -2 >> $o
PXA_OPERAND_DTYPE=f16 timeout 1500 python -m pytest tests/test_model_gpu.py -x -q 2>&1 | tail -2 >> $o
PXA_ATTN_NO_KVRES=1 timeout 300 python tools/kbench_cross.py 2>&1 | grep "L=" | sed "s/^/streaming: /" >> $o
timeout 300 python tools/kbench_cross.py 2>&1 | grep "L=" | sed "s/^/resident : /" >> $o
for v in resident streaming resident streaming; do
  if [ $v = streaming ]; then export PXA_ATTN_NO_KVRES=1; else unset PXA_ATTN_NO_KVRES; fi
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['dtype'], round(d['ms_per_step'],1), 'ms')" >> $o
done
cat $o
