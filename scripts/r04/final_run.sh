#!/bin/bash
# round-4 evidence, all at the stamped commit (.kf_commit): the headline pin (bench lines, clocks, rocprofv3 kernel stats, HBM and SQ
# counters), kernel stats + HBM traffic of C2 / C3 / C4, bench lines of every BASELINE workload, the strong-scaling proxy table, the
# time-varying / dense-H lines, logp + gradient kernel stats and sweep, the full GPU test suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r04_final; mkdir -p $O
scripts/r04/headline_pin.sh r04_pin_final > $O/headline_pin.txt 2>&1
cp gpurun_out/r04_pin_final/headline_summary.txt $O/headline_summary.txt
cp gpurun_out/r04_pin_final/clock_ramp.txt $O/clock_ramp.txt
cp gpurun_out/r04_pin_final/traffic_entry.json $O/traffic_headline.json 2>/dev/null
for w in c2 c3 c4; do
  scripts/profile.sh r04_$w --workload $w > $O/${w}_summary.txt 2>&1
  cp gpurun_out/r04_$w/traffic_entry.json $O/traffic_$w.json 2>/dev/null
done
for w in headline c2 c3 c4 c5; do
  steps=20; [ $w = c5 ] && steps=2
  python3 bench.py --workload $w --steps $steps --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
scripts/r04/strong_proxy.sh r04_strong_final > $O/strong_proxy.log 2>&1
cp gpurun_out/r04_strong_final/strong_proxy.md $O/strong_proxy.md
scripts/r04/tv_dense.sh > $O/tv_dense_lines.txt 2>&1
python3 scripts/r04/vg_time.py > $O/vg_time.txt 2>&1
python3 scripts/r04/vg_time.py sweep > $O/vg_sweep.txt 2>&1
python3 scripts/r04/vg_time.py hint > $O/vg_placement_hint.txt 2>&1
scripts/r04/vg_prof.sh > $O/vg_kernel_stats.txt 2>&1
python3 scripts/r04/smooth_split_check.py c3 > $O/smooth_split_c3.txt 2>&1
KF_SMOOTHER_SPLIT=1 python3 scripts/r04/smooth_split_check.py headline > $O/smooth_split_headline_forced.txt 2>&1
( cd /tmp; export TMPDIR=/tmp; for w in headline c3; do
    KF_SMOOTHER_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r04_smooth/prof_$w -- python3 $ROOT/scripts/r04/smooth_prof.py $w > /dev/null 2>&1
    echo "== $w (KF_SMOOTHER_SPLIT=1)"; python3 $ROOT/scripts/kernel_stats.py $ROOT/gpurun_out/r04_smooth/prof_$w 7
  done ) > $O/smooth_split_kernel_stats.txt 2>&1
scripts/r04/c3_run.sh > $O/c3_run.txt 2>&1
cp gpurun_out/r04_c3/c3_split_timeline.txt $O/c3_split_timeline.txt
python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -3 $O/pytest_gpu.txt
cat .kf_commit
find gpurun_out -name "*.csv" -size +1M -delete
