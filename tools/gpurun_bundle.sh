#!/bin/bash
# Build container side of a validation bundle: records the HEAD the snapshot carries, ships it, then copies the judged artefacts into profiles/.
#   usage (repo root, clean tree):  bash tools/gpurun_bundle.sh [tag]
tag=${1:-r5final}
cd "$(dirname "$0")/.." || exit 1
echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet || echo +dirty)" > .gpurun_head
/usr/local/graft/bin/gpurun --timeout 3000 -- "bash tools/sessions/r5_bundle.sh $tag" > /tmp/gpurun_bundle_$tag.log 2>&1
tail -40 /tmp/gpurun_bundle_$tag.log
bash tools/collect_bundle.sh $tag
