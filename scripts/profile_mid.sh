#!/bin/bash
# kernel stats of forward + gradient + smoother at mid-size states (block-tile and workgroup kernels)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
for M in 32 64; do
  OUT=$ROOT/gpurun_out/r01_mid_m$M
  mkdir -p "$OUT"
  cd /tmp
  NB_WHAT=fwd,smooth,grad NB_M=$M NB_N=200 NB_B=4096 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 $ROOT/scripts/probes/big_mid.py > "$OUT/log.txt" 2>&1
  cd "$ROOT"
  python3 - "$OUT" $M <<'PY'
import csv, glob, os, sys
d, m = sys.argv[1], sys.argv[2]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(f"## m = {m}, 4096 draws x 200 steps, p = 1, r = m/3: rocprofv3 --kernel-trace --stats")
    print(f"{'kernel':72s} {'calls':>6s} {'avg_ms':>10s} {'pct':>6s}")
    for r in rows[:6]:
        print(f"{r['Name'][:72]:72s} {r['Calls']:>6s} {float(r['AverageNs'])/1e6:10.3f} {float(r['Percentage']):6.2f}")
PY
  find "$OUT" -name "*kernel_trace.csv" -delete
done
