#!/bin/bash
# usage: scripts/profile_ops.sh <tag> [workload]   -- kernel stats of forward + gradient + smoother
set -u
TAG=$1; WL=${2:-headline}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/scripts/profile_ops.py $WL > "$OUT/log.txt" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    print("## rocprofv3 --kernel-trace --stats:", os.path.basename(f))
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':70s} {'calls':>6s} {'avg_us':>10s} {'pct':>6s}")
    for r in rows[:12]:
        print(f"{r['Name'][:70]:70s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")
PY
find "$OUT" -name "*kernel_trace.csv" -size +1M -delete
