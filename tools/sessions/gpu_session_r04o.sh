cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04o_kvlin_step.txt
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "train" 2>&1 | tail -2 > $o
PXA_OPERAND_DTYPE=f16 timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "train" 2>&1 | tail -2 >> $o
cat gpurun_out/r04n_kvlin.txt >> $o
cat $o
