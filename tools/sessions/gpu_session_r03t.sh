# GPU-box session r03t: 4-wave GEMM (128 x 128 per wave, accumulators in AGPRs, BK = 64, compiler-scheduled 2-stage loop) vs the persistent kernel vs hipBLASLt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out/r03t_gemm_w4.txt
: > $o
for shape in "65536 4608 1152" "65536 1152 4608" "65536 3456 1152"; do
 for lay in NT NN; do
  echo "== $lay $shape" >> $o
  timeout 120 python tools/kbench_one.py $shape $lay 20 2>&1 | grep TF | sed "s/^/pers        : /" >> $o
  KBENCH_LIBREF=1 timeout 120 python tools/kbench_one.py $shape $lay 20 2>&1 | grep TF | sed "s/^/hipBLASLt   : /" >> $o
  PXA_GEMM_NO_PERSISTENT=1 timeout 120 python tools/kbench_one.py $shape $lay 20 2>&1 | grep TF | sed "s/^/glds 8 wave : /" >> $o
  PXA_LIB_PATH=$GRAFT_REPO_ROOT/pixart_sigma_amd/variants/lib_w4.so PXA_GEMM_TILE=2564 timeout 120 python tools/kbench_one.py $shape $lay 20 2>&1 | grep TF | sed "s/^/glds 4 wave : /" >> $o
 done
done
cat $o
