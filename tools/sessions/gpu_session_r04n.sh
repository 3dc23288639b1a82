cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 200 python tools/kbench_kvlin.py 2>&1 | grep "NN dX" > gpurun_out/r04n_kvlin.txt
cat gpurun_out/r04n_kvlin.txt
