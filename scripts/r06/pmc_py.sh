#!/bin/bash
# SQ counter passes (one rocprofv3 --pmc run per group, kernel trace only) of one python command; per-kernel means per launch.
#   usage: scripts/r06/pmc_py.sh <tag> <script.py> [args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/$1" "${@:2}" > /dev/null 2> "$OUT/pmc$i.log"
done
cd "$ROOT"
python3 - "$OUT" <<'PY' | tee "$OUT/counters.txt"
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "kf_" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean/launch = {sum(v)/len(v):.5g}  (n={len(v)})")
PY
find "$OUT" -name "*.csv" -size +1M -delete
