# GPU-box session r04l: keys-resident dQ kernel with the next trip's rows prefetched
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04l_kvres_dq_prefetch.txt
timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "attention" 2>&1 | tail -2 > $o
timeout 300 python tools/kbench_cross.py 2>&1 | grep "L=" | sed "s/^/resident + prefetch: /" >> $o
timeout 300 python tools/kbench_cross.py 2>&1 | grep "L= 300" | sed "s/^/resident + prefetch: /" >> $o
cat $o
