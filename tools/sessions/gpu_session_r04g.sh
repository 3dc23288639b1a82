# GPU-box session r04g: HBM-bound row kernels beside a persistent GEMM on a second stream
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/overlap_streams.py 2>&1 | grep "GEMM alone" > gpurun_out/r04g_overlap_streams.txt
cat gpurun_out/r04g_overlap_streams.txt
