#!/bin/bash
# gpurun wrapper: stamps the tree with the commit it was built from (the GPU box has no .git) so that
# profiles/ summaries and traffic.json entries name the commit they were measured at.
#   usage: scripts/grun.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.." || exit 1
stamp=$(git rev-parse --short HEAD)
git diff --quiet HEAD -- pymc_statespace_amd include bench.py || stamp="$stamp+dirty"
echo "$stamp" > .kf_commit
exec /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"
