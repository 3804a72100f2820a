#!/bin/bash
# copy the evidence of scripts/r04/final_run.sh (gpurun_out/r04_final) into profiles/r04/ under fixed names; COMMIT.txt = the commit the
# lease ran at (files of earlier leases of the round: profiles/r04/earlier/, their commit in the name)
cd "$(dirname "$0")/../.."; F=gpurun_out/r04_final; P=profiles/r04
for f in headline_pin headline_summary clock_ramp c2_summary c3_summary c4_summary tv_dense_lines vg_time vg_sweep vg_placement_hint \
         vg_kernel_stats smooth_split_c3 smooth_split_headline_forced smooth_split_kernel_stats c3_split_timeline pytest_gpu; do cp $F/$f.txt $P/$f.txt; done
cp $F/strong_proxy.md $P/strong_proxy.md
for w in headline c2 c3 c4 c5; do cp $F/bench_$w.json $P/bench_$w.json; done
python3 scripts/merge_traffic.py $F/traffic_*.json > /dev/null
tail -1 gpurun_out/final_run.log > $P/COMMIT.txt
cat $P/COMMIT.txt; tail -2 $P/pytest_gpu.txt
