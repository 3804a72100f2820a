#!/bin/bash
# rocprofv3 kernel trace + stats of one python command; prints the per-kernel table (name, calls, avg us, total us).
#   usage: scripts/r06/kt.sh <tag> <script.py> [args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 "$ROOT/$1" "${@:2}" > "$OUT/stdout.txt" 2> "$OUT/kt.log"
cd "$ROOT"
F=$(find "$OUT/kt" -name "*kernel_stats.csv" | head -1)
if [ -n "$F" ]; then
  python3 - "$F" <<'PY' | tee "$OUT/kernel_stats.txt"
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(f"{'kernel':90s} {'calls':>6s} {'avg us':>10s} {'total ms':>10s} {'%':>6s}")
for r in rows[:14]:
    print(f"{r['Name'][:90]:90s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:10.2f} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['Percentage']):6.2f}")
PY
else
  echo "no kernel_stats.csv"; tail -5 "$OUT/kt.log"
fi
find "$OUT" -name "*.csv" -size +2M -delete
find "$OUT" -name "*.db" -delete
