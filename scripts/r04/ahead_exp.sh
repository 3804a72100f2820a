#!/bin/bash
# kf_fwd_b4_kernel in the steady-state value-and-gradient path: with its convergence test (steady), with the test never running -- a missing
# last observation (steady_nan) --, and the plain instantiation (exact): kernel stats at 4096 and 512 draws
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r04_ahead; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in headline h8; do for x in steady steady_nan exact; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$w$x -- python3 $ROOT/scripts/r04/vg_prof.py $w $x > /dev/null 2>&1
  echo "== $w $x"; python3 $ROOT/scripts/kernel_stats.py $O/$w$x 4
done; done
find $O -name "*kernel_trace.csv" -size +1M -delete
