#!/bin/bash
# GPU-box profiling recipe (rocprofv3): per-kernel time, then HBM traffic counters in their
# own passes (TCC slots: FETCH_SIZE and WRITE_SIZE do not fit one pass; never mix --pmc with
# tracing domains other than kernel-trace).
#   usage: scripts/profile.sh <tag> [bench.py args...]
set -u
TAG=${1:-prof}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
BENCH="$ROOT/bench.py --steps 10 --warmup 2 --no-cpu --no-ops $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $BENCH > "$OUT/bench_kt.json" 2> "$OUT/kt.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $BENCH > "$OUT/bench_write.json" 2> "$OUT/write.log"
cd "$ROOT"
python3 scripts/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
# keep the merged-back directory small
find "$OUT" -name "*.csv" -size +2M -delete
