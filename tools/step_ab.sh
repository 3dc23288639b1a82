#!/bin/bash
# In-step A/B on ONE box: the training step (bench.py, fp16 build, 8 steps after 3 of warm-up) under alternating configurations, two rounds.  The stand-alone
# kernel benches decide nothing by themselves any more (two of their verdicts were reversed by this measurement in round 4).
# A change that touches a kernel EVERY configuration runs needs the previous commit's library as a partner ("old|PXA_LIB_PATH=..." built with tools/build_variant.py
# from `git show HEAD~1:...`), not an environment switch inside the new library: round 5's ln_mod_bwd change slowed all 57 calls of the kernel by 140 us and its own
# A/B - new kernel with / without the new pointer - showed 0.5 ms in its favour.
#   usage (GPU box, repo root):  bash tools/step_ab.sh OUT.txt "label1|ENV1=a ENV2=b" "label2|PXA_LIB_PATH=pixart_sigma_amd/variants/lib_x.so" ...
#   a configuration with no environment: "default|"
out=$1; shift
export PYTHONUNBUFFERED=1
: > "$out"
for rep in 1 2; do
  for cfg in "$@"; do
    label=${cfg%%|*}; envs=${cfg#*|}
    r=$(env $envs timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-roofline --no-other-dtype --no-torch-baseline --no-configs 2>/dev/null \
        | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["final_loss"])')
    echo "$label: $r" >> "$out"
  done
done
cat "$out"
