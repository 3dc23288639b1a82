#!/bin/bash
# round 4, session 3: where a tile's time goes in attn_fwd4_kernel - s_memtime sums per section (barrier wait / phase A / phase B), with ablations
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out
mkdir -p $O
export PYTHONUNBUFFERED=1
for v in trace tracef trace_abl1 trace_abl2 trace_abl4 trace_abl8 trace_abl16 trace_abl24 trace_abl39; do
  PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f4_$v.so timeout 120 python tools/kbench_fwd4.py trace 2>&1 | grep "^trace" >> $O/r4_03_fwd4_trace.txt
  PXA_LIB_PATH=pixart_sigma_amd/variants/lib_f4_$v.so timeout 120 python tools/kbench_fwd4.py time 2>&1 | grep "FWD4=1" | tail -1 >> $O/r4_03_fwd4_trace.txt
done
cat $O/r4_03_fwd4_trace.txt
