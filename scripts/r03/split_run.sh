#!/bin/bash
# split pipeline: parity check + timing + kernel timeline of the last pass (rocprofv3 --kernel-trace)
#   usage (through gpurun): scripts/r03/split_run.sh [small] [workload ...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
export TMPDIR=/tmp
if [ "$1" = "small" ]; then shift; python3 scripts/r03/split_check.py small > gpurun_out/split_small.txt 2>&1; tail -3 gpurun_out/split_small.txt; fi
python3 scripts/r03/split_check.py headline > gpurun_out/split_headline.txt 2>&1; grep -v amdgpu.ids gpurun_out/split_headline.txt
for w in ${@:-headline}; do
  ( cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_${w}_full -- python3 $R/scripts/r03/split_trace.py $w full > $R/gpurun_out/trace_${w}_full.log 2>&1 )
  python3 scripts/r03/trace_timeline.py gpurun_out/trace_${w}_full > gpurun_out/timeline_${w}_full.txt; cat gpurun_out/timeline_${w}_full.txt
done
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete
