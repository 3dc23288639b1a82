#!/bin/bash
# Build-container side of a GPU session: records the HEAD the snapshot carries (.gpurun_head; "+dirty" when the tree has uncommitted changes), ships it and runs
# tools/session.sh on the box.    usage (repo root):  bash tools/gpurun_session.sh TIMEOUT_S TAG recipe [recipe ...]
cd "$(dirname "$0")/.." || exit 1
to=$1; tag=$2; shift 2
echo "$(git rev-parse --short=12 HEAD)$(git diff --quiet || echo +dirty)" > .gpurun_head
q=""
for a in "$@"; do q="$q '$a'"; done
/usr/local/graft/bin/gpurun --timeout "$to" -- "bash tools/session.sh $tag $q" > /tmp/gpurun_$tag.log 2>&1
echo "gpurun rc=$?" >> /tmp/gpurun_$tag.log
tail -n 80 /tmp/gpurun_$tag.log
