#!/bin/bash
# SQ counters of the smoother kernel at the headline shape, one rocprofv3 --pmc pass per counter group
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/spmc$i
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/spmc$i -- python3 $R/scripts/r03/smooth_only.py > /dev/null 2> $R/gpurun_out/spmc$i.log
done
cd $R
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/spmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "smooth" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean/launch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
find gpurun_out -name "*.csv" -size +1M -delete
