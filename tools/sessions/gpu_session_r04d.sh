# GPU-box session r04d: cross-attention kernel time vs text length (per-tile work vs per-workgroup overhead)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/kbench_cross.py 2>&1 | grep "L=" > gpurun_out/r04d_cross_vs_len.txt
cat gpurun_out/r04d_cross_vs_len.txt
