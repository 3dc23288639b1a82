# GPU-box session: final attention build (no SLP vectorisation, delta folded into dP in the dQ kernel only): parity + timing
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
timeout 40 python -m pytest tests/test_kernels_gpu.py -q -k "attention" > $o/r02o_pytest_attention.log 2>&1; echo "pytest rc $?" >> $o/r02o_pytest_attention.log
{ timeout 30 python tools/kbench_attn_bwd.py; timeout 30 python tools/kbench.py attn; } 2>&1 | grep -v amdgpu.ids > $o/r02o_attn_final.txt
tail -3 $o/r02o_pytest_attention.log; cat $o/r02o_attn_final.txt
