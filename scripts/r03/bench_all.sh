#!/bin/bash
# every BASELINE workload through bench.py at the stamped commit -> gpurun_out/r03_bench/bench_<workload>.json
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/r03_bench${1:+_$1}
mkdir -p $O
nproc > $O/nproc.txt; cat /sys/fs/cgroup/cpu.max >> $O/nproc.txt 2>&1; python3 -c "import os; print(len(os.sched_getaffinity(0)), os.cpu_count())" >> $O/nproc.txt
for w in headline c2 c3 c4 c5; do
  steps=20; [ $w = c5 ] && steps=2
  python3 bench.py --workload $w --steps $steps --warmup 2 > $O/bench_$w.json 2> $O/bench_$w.err
  echo "$w rc=$?"; tail -c 600 $O/bench_$w.json
done
python3 bench.py --workload headline --batch 512 --no-cpu > $O/bench_headline_b512.json 2> $O/bench_headline_b512.err
python3 bench.py --workload c4 --batch 2048 --no-cpu > $O/bench_c4_b2048.json 2> $O/bench_c4_b2048.err
cat .kf_commit
