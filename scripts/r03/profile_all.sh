#!/bin/bash
# round-3 evidence, all at the stamped commit: bench lines of every BASELINE workload (+ the one-of-eight-GPUs shards), rocprofv3
# kernel stats + HBM traffic counters of the headline / C3 / C4 passes, SQ counters of the headline kernel, kernel stats of
# logp + gradient (C4) and of the smoothers.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/r03_prof
mkdir -p $O
scripts/r03/bench_all.sh final > $O/bench_all.log 2>&1
scripts/profile.sh r03_headline > $O/headline_summary.txt 2>&1
scripts/profile.sh r03_c3 --workload c3 > $O/c3_summary.txt 2>&1
scripts/profile.sh r03_c4 --workload c4 > $O/c4_summary.txt 2>&1
scripts/profile.sh r03_c2 --workload c2 > $O/c2_summary.txt 2>&1
scripts/pmc.sh r03_sq1 "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" --no-ops > $O/headline_sq1.txt 2>&1
scripts/pmc.sh r03_sq2 "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" --no-ops > $O/headline_sq2.txt 2>&1
export TMPDIR=/tmp
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_vg_c4 -- python3 $ROOT/scripts/r03/vg_trace.py 16384 > $ROOT/$O/vg_c4.log 2>&1 )
python3 scripts/kernel_stats.py gpurun_out/r03_vg_c4 8 > $O/vg_c4_kernel_stats.txt
( cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/r03_trace_c3 -- python3 $ROOT/scripts/r03/split_trace.py c3 full > $ROOT/$O/trace_c3.log 2>&1 )
python3 scripts/r03/trace_timeline.py gpurun_out/r03_trace_c3 > $O/c3_split_timeline.txt
# the C5 share: fused three-launch step against the separate kernels (KF_BIG_FUSE=0), per-kernel averages, the C5 tests
scripts/r03/c5_run.sh > $O/c5_fused_ab.txt 2>&1
# 64 < m <= 96 on chip (kf_forward_wgl.hip) against the workspace path; m = 128 stays on the latter
python3 scripts/r03/wgl_check.py > $O/wgl_check.txt 2>&1
python3 scripts/r03/mid_m_bench.py > $O/mid_m_bench.txt 2>&1
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_wgl -- python3 $ROOT/scripts/r03/wgl_time.py > $ROOT/$O/wgl_time.txt 2>&1 )
python3 scripts/kernel_stats.py gpurun_out/r03_wgl 6 > $O/wgl_kernel_stats.txt
# smoothers: gain reuse against every step in full, the constant-tape / slow-draw experiment, kernel stats at the headline shape
python3 scripts/r03/smooth_steady_check.py > $O/smooth_steady_check.txt 2>&1
python3 scripts/r03/smooth_lane_steady_check.py > $O/smooth_lane_steady_check.txt 2>&1
python3 scripts/r03/smooth_const_tape.py > $O/smooth_const_tape.txt 2>&1
( cd /tmp; rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/r03_smooth -- python3 $ROOT/scripts/r03/smooth_only.py > /dev/null 2>&1 )
python3 scripts/kernel_stats.py gpurun_out/r03_smooth 6 > $O/smoother_kernel_stats.txt
for d in r03_headline r03_c3 r03_c4 r03_c2; do cp gpurun_out/$d/traffic_entry.json $O/traffic_$d.json 2>/dev/null; done
find gpurun_out -name "*kernel_trace.csv" -size +1M -delete
find gpurun_out -name "*counter_collection.csv" -size +1M -delete
cat $O/headline_summary.txt | tail -30
