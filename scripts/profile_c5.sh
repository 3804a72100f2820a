#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r01_c5
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/scripts/gpu_c5.py 128 32 > "$OUT/log.txt" 2>&1
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys
d = sys.argv[1]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f"{'kernel':80s} {'calls':>6s} {'avg_us':>10s} {'total_ms':>9s} {'pct':>6s}")
    for r in rows[:22]:
        print(f"{r['Name'][:80]:80s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['TotalDurationNs'])/1e6:9.2f} {float(r['Percentage']):6.2f}")
PY
tail -3 "$OUT/log.txt"
find "$OUT" -name "*kernel_trace.csv" -size +1M -delete
