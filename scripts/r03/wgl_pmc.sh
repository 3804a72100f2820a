#!/bin/bash
# SQ counters of the LDS-resident forward kernels (kf_forward_wgl.hip), one rocprofv3 --pmc pass per counter group
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
i=0
for CTRS in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/wpmc$i
  rocprofv3 --kernel-trace --pmc $CTRS --output-format csv -d $R/gpurun_out/wpmc$i -- python3 $R/scripts/r03/wgl_time.py > /dev/null 2> $R/gpurun_out/wpmc$i.log
done
cd $R
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/wpmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "wgl" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean/launch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
find gpurun_out -name "*.csv" -size +1M -delete
