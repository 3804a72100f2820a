#!/bin/bash
# HBM traffic + kernel stats of C3 / C4 (and the headline) at the stamped commit -> gpurun_out/r05_traffic
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$ROOT"; O=$ROOT/gpurun_out/r05_traffic; mkdir -p $O
for w in headline c3 c4; do
  scripts/profile.sh r05_$w --workload $w > $O/${w}_summary.txt 2>&1
  cp gpurun_out/r05_$w/traffic_entry.json $O/traffic_$w.json 2>/dev/null
  grep -E "whole pass|traffic entry|algorithmic" $O/${w}_summary.txt | cut -c1-400
done
find gpurun_out -name "*.csv" -size +1M -delete
