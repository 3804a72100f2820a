#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd /tmp; export TMPDIR=/tmp
for w in headline h8 c3; do for x in steady exact; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_vg_$w$x -- python3 $ROOT/scripts/r04/vg_prof.py $w $x > /dev/null 2>&1
  echo "== $w $x"; python3 $ROOT/scripts/kernel_stats.py $ROOT/gpurun_out/r04_vg_$w$x 8
done; done
find $ROOT/gpurun_out -name "*kernel_trace.csv" -size +1M -delete
