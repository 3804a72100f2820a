#!/bin/bash
# SQ counters of the headline-shape op kernels (forward, adjoint sweeps, smoother): two rocprofv3 --pmc passes over scripts/profile_ops.py
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
run() {
  OUT=$ROOT/gpurun_out/r02_pmc_ops_$1; mkdir -p "$OUT"
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $2 --output-format csv -d "$OUT" -- python3 $ROOT/scripts/profile_ops.py headline > "$OUT/log.txt" 2>&1 )
  python3 - "$OUT" <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "kf_" not in k or "fill" in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean/launch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
  find "$OUT" -name "*.csv" -size +1M -delete
}
run A "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS"
run B "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
