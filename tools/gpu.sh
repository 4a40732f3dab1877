#!/bin/bash
# gpurun wrapper: records the commit the snapshot is built from (the GPU box has no .git; tools/pmc_*summary.py copy it into the
# profile files, bench.py reports it as roofline.traffic_source_commit), then runs the command on an MI355X box.
#   tools/gpu.sh [--timeout S] -- '<command>'
cd "$(dirname "$0")/.."
c=$(git rev-parse --short=12 HEAD 2>/dev/null)
git diff --quiet HEAD -- zksnark_rs_amd include bench.py 2>/dev/null || c="$c+uncommitted"
echo "$c" > .build_commit
exec /usr/local/graft/bin/gpurun "$@"
