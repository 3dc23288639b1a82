# GPU-box session r04e: cross-attention backward with each dQ / dK,dV kernel generation (the hand-placed round-3 kernels have longer prologues)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r04e_cross_kernel_modes.txt
: > $o
for dq in 0 1; do for dkv in 0 1 2; do
  PXA_ATTN_DQ=$dq PXA_ATTN_DKV=$dkv timeout 300 python tools/kbench_cross.py 2>&1 | grep "L= 300\|L= 128" | sed "s/^/dq=$dq dkv=$dkv: /" >> $o
done; done
cat $o
