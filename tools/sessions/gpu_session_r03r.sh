# GPU-box session r03r: ablation of the forward attention loop (FWD_ABL 1..4)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
for v in default fabl1 fabl2 fabl3 fabl4 default; do
  if [ $v = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$GRAFT_REPO_ROOT/pixart_sigma_amd/variants/lib_$v.so; fi
  timeout 300 python tools/kbench.py attn 2>&1 | grep "attn fwd" | sed "s/^/$v: /"
done > $o/r03r_fwd_ablation.txt
cat $o/r03r_fwd_ablation.txt
