#!/bin/bash
# Build and run the issue-rate probe (tools/probes/issue_probe.cpp: a standalone program, NOT part of any library build) on the GPU box.
#   bash tools/issue_probe.sh [workgroups = 256]  ->  JSON lines on stdout (profiles/r5_issue_probe.jsonl is one such run)
# Build here (no GPU needed):  bash tools/issue_probe.sh --build-only
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/tools/_build
BIN=$R/tools/_build/issue_probe
if [ ! -x $BIN ] || [ $R/tools/probes/issue_probe.cpp -nt $BIN ]; then
  ${HIPCC:-/opt/rocm/bin/hipcc} -O3 -std=c++17 --offload-arch=gfx950 -x hip $R/tools/probes/issue_probe.cpp -o $BIN
fi
[ "$1" == "--build-only" ] && exit 0
exec $BIN "$@"
