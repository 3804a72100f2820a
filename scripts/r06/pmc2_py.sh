#!/bin/bash
# Second set of SQ counter passes (which pipe is busy: LDS array, MFMA, scalar, VMEM) -- same conventions as pmc_py.sh.
#   usage: scripts/r06/pmc2_py.sh <tag> <script.py> [args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
i=0
for c in "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INST_CYCLES_SALU" "SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_IFETCH" "SQ_WAIT_INST_ANY SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/pmc$i" -- python3 "$ROOT/$1" "${@:2}" > /dev/null 2> "$OUT/pmc$i.log"
  tail -2 "$OUT/pmc$i.log" | grep -i -E "error|invalid|not" | head -2
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
