#!/bin/bash
# round 4, session 40: static and dynamic item hand-out of the persistent GEMMs give bit-identical outputs
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 60 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -p no:cacheprovider -k "item_schedulers" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r4_40_schedulers_agree.txt
