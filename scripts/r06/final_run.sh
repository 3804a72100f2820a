#!/bin/bash
# round-6 evidence at the stamped commit (.kf_commit): the headline pin (bench lines, clocks, rocprofv3 kernel stats, HBM counters),
# kernel stats + HBM traffic of the chunked pipeline at the headline's one-of-eight share and of C2, bench lines of the BASELINE
# workloads and their shares, the strong-scaling proxy table.   (the full GPU suite: its own lease, profiles/r06/pytest_gpu_*.txt)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r06_final; mkdir -p $O
scripts/r04/headline_pin.sh r06_pin > $O/headline_pin.txt 2>&1
cp gpurun_out/r06_pin/headline_summary.txt $O/headline_summary.txt 2>/dev/null
cp gpurun_out/r06_pin/traffic_entry.json $O/traffic_headline.json 2>/dev/null
scripts/profile.sh r06_h512 --workload headline --batch 512 > $O/headline_b512_summary.txt 2>&1
cp gpurun_out/r06_h512/traffic_entry.json $O/traffic_headline_b512.json 2>/dev/null
scripts/profile.sh r06_c2 --workload c2 > $O/c2_summary.txt 2>&1
cp gpurun_out/r06_c2/traffic_entry.json $O/traffic_c2.json 2>/dev/null
for w in headline c2 c3 c4; do
  python3 bench.py --workload $w --steps 20 --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
python3 bench.py --workload headline --batch 512 --steps 20 --warmup 5 > $O/bench_headline_b512.json 2> $O/bench_headline_b512.err
python3 bench.py --workload headline --batch 256 --steps 20 --warmup 5 > $O/bench_headline_b256.json 2> $O/bench_headline_b256.err
scripts/r04/strong_proxy.sh r06_strong > $O/strong_proxy.log 2>&1
cp gpurun_out/r06_strong/strong_proxy.md $O/strong_proxy.md
python3 scripts/r06/smooth_lane_time.py > $O/smooth_lane_time.txt 2>&1
python3 scripts/r06/chunk_check.py > $O/chunk_check.txt 2>&1
cat .kf_commit > $O/COMMIT.txt
ls $O
