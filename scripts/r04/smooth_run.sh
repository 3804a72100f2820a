#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r04_smooth; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_r04.py -x -q -k "smoother" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
KF_SMOOTHER_SPLIT=1 python3 scripts/r04/smooth_split_check.py 2>&1 | grep -v amdgpu.ids
cd /tmp; export TMPDIR=/tmp
for w in headline c3; do
  KF_SMOOTHER_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 $ROOT/scripts/r04/smooth_prof.py $w > /dev/null 2>&1
  echo "== $w"; python3 $ROOT/scripts/kernel_stats.py $O/prof_$w 6
done
find $O -name "*kernel_trace.csv" -size +1M -delete
