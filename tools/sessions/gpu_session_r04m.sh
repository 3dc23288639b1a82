# GPU-box session r04m: delta pre-pass folded into the hand-placed dQ kernel (self-attention): parity, all 256 heads at B16, attention time, step A/B
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04m_dq_fold.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -2 > $o
timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "train or checkpoint or block" 2>&1 | tail -2 >> $o
PXA_OPERAND_DTYPE=f16 timeout 1500 python -m pytest tests/test_model_gpu.py -x -q -k "train or checkpoint or block" 2>&1 | tail -2 >> $o
for v in fold nofold fold nofold; do
  if [ $v = nofold ]; then export PXA_ATTN_DQ_FOLD=0; else unset PXA_ATTN_DQ_FOLD; fi
  timeout 300 python tools/kbench.py attn 2>&1 | grep "attn bwd self" | sed "s/^/$v: /" >> $o
done
for v in fold nofold fold nofold; do
  if [ $v = nofold ]; then export PXA_ATTN_DQ_FOLD=0; else unset PXA_ATTN_DQ_FOLD; fi
  timeout 300 python bench.py --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['dtype'], round(d['ms_per_step'],1), 'ms')" >> $o
done
cat $o
