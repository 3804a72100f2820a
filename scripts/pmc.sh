#!/bin/bash
# usage: scripts/pmc.sh <tag> "<counters>" [bench args...]   -- one rocprofv3 --pmc pass
set -u
TAG=$1; CTRS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d "$OUT" -- python3 $ROOT/bench.py --steps 4 --warmup 1 --no-cpu "$@" > "$OUT/bench.json" 2> "$OUT/log.txt"
cd "$ROOT"
python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "kf_" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean/launch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
find "$OUT" -name "*.csv" -size +1M -delete
