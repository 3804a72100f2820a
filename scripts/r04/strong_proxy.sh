#!/bin/bash
# the strong-scaling PROXY of every 8-GPU configuration, from bench lines -- the whole batch on one
# GPU next to one of eight GPUs' share on one GPU (no 8-GPU node is ours to launch: the gain available from sharding the draw
# axis is single-GPU time(B) / single-GPU time(B / 8); the all-gather of B / 8 doubles per rank is microseconds).
#   usage: scripts/r04/strong_proxy.sh [tag]   -> gpurun_out/<tag>/*.json, strong_proxy.md
set -u
TAG=${1:-r04_strong}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=gpurun_out/$TAG; mkdir -p $O
run() {  # run <name> <bench args...>
  local name=$1; shift
  python3 bench.py --no-cpu "$@" > $O/$name.json 2> $O/$name.err
  echo "$name rc=$? $(tail -c 200 $O/$name.json | head -c 0)"
}
run headline_b4096 --workload headline --steps 20 --warmup 5
run headline_b512  --workload headline --batch 512 --steps 20 --warmup 5
run c3_b8192       --workload c3 --steps 10 --warmup 3
run c3_b1024       --workload c3 --batch 1024 --steps 10 --warmup 3
run c4_b16384      --workload c4 --steps 10 --warmup 3
run c4_b2048       --workload c4 --batch 2048 --steps 10 --warmup 3
run c5_b128        --workload c5 --batch 128 --steps 2 --warmup 1
run c5_b1024       --workload c5 --batch 1024 --steps 1 --warmup 1
python3 scripts/r04/strong_proxy_table.py $O > $O/strong_proxy.md
cat $O/strong_proxy.md
