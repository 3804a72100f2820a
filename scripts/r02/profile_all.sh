#!/bin/bash
# round-2 profiles: headline (kernel stats + HBM traffic counters), C5 (kernel stats), C5 / op / distributed tests.
#   usage (through gpurun): KF_COMMIT=$(git rev-parse --short HEAD) scripts/r02/profile_all.sh
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/r02_prof
scripts/profile.sh r02_headline > gpurun_out/r02_prof/headline_summary.txt 2>&1
python3 bench.py --workload c5 --steps 2 --warmup 1 > gpurun_out/r02_prof/bench_c5.json 2> gpurun_out/r02_prof/bench_c5.err
export TMPDIR=/tmp
OUT=$ROOT/gpurun_out/r02_c5
mkdir -p "$OUT"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --workload c5 --steps 1 --warmup 1 --no-cpu > "$OUT/bench.json" 2> "$OUT/log.txt" )
python3 - "$OUT" > gpurun_out/r02_prof/c5_kernel_stats.txt <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':84s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'pct':>6s}")
    for r in rows[:16]:
        print(f"{r['Name'][:84]:84s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):6.2f}")
PY
for f in $(find "$OUT" -name "*kernel_stats.csv"); do cp "$f" gpurun_out/r02_prof/c5_kernel_stats.csv; done
find "$OUT" -name "*kernel_trace.csv" -size +1M -delete
cat gpurun_out/r02_prof/headline_summary.txt gpurun_out/r02_prof/bench_c5.json gpurun_out/r02_prof/c5_kernel_stats.txt
