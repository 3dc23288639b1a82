# GPU-box session r03z: persistent GEMM with staggered workgroup start phases (epilogue store bursts de-synchronised) vs lock-step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r03z_gemm_stagger.txt
: > $o
for rep in 1 2; do
for v in default stag200 stag450; do
  if [ $v = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=$GRAFT_REPO_ROOT/pixart_sigma_amd/variants/lib_$v.so; fi
  timeout 300 python tools/kbench_gelu.py 2>&1 | grep "M= 65536" | sed "s/^/$v: /" >> $o
  timeout 120 python tools/kbench_one.py 65536 1152 4608 NT 20 2>&1 | grep TF | sed "s/^/$v: /" >> $o
  timeout 120 python tools/kbench_one.py 65536 4608 1152 NN 20 2>&1 | grep TF | sed "s/^/$v: /" >> $o
done; done
cat $o
