#!/bin/bash
# copy the evidence of scripts/r05/final_run.sh (gpurun_out/r05_final) and scripts/r05/pytest_full.sh (gpurun_out/r05_full) into
# profiles/r05/ under fixed names; COMMIT.txt = the commit the leases ran at
cd "$(dirname "$0")/../.."; F=gpurun_out/r05_final; P=profiles/r05
for f in headline_pin headline_summary clock_ramp c2_summary c3_summary c4_summary c5_summary tv_lines denseh_lines vg_time \
         smooth_from_filter_time smooth_from_filter_prof c5_smoother_from_filter c5_smoother_from_filter_kernels c5_sparse_step_counters \
         c5_value_and_grad_kernels tv_gradient_smoother_time denseh_same_data; do
  [ -f $F/$f.txt ] && cp $F/$f.txt $P/$f.txt
done
cp $F/strong_proxy.md $P/strong_proxy.md
for w in headline c2 c3 c4 c5; do cp $F/bench_$w.json $P/bench_$w.json; done
python3 scripts/merge_traffic.py $F/traffic_*.json > /dev/null
[ -f gpurun_out/r05_full/pytest_gpu.txt ] && cp gpurun_out/r05_full/pytest_gpu.txt $P/pytest_gpu.txt
[ -f gpurun_out/r05_full/fuzz_summary.txt ] && cp gpurun_out/r05_full/fuzz_summary.txt $P/fuzz_summary.txt
cat .kf_commit > $P/COMMIT.txt
cat $P/COMMIT.txt; tail -2 $P/pytest_gpu.txt
