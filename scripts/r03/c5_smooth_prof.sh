#!/bin/bash
R=$GRAFT_REPO_ROOT
export TMPDIR=/tmp; cd /tmp
rm -rf $R/gpurun_out/c5s
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c5s -- python3 $R/scripts/r03/c5_smooth_prof.py 2>/dev/null | tail -2
cd $R; python3 scripts/kernel_stats.py gpurun_out/c5s 25
