#!/bin/bash
# HBM traffic of the LDS-resident forward kernels (FETCH_SIZE / WRITE_SIZE in separate rocprofv3 --pmc passes, as scripts/profile.sh)
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/wtr_$C
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/gpurun_out/wtr_$C -- python3 $R/scripts/r03/wgl_time.py > /dev/null 2> $R/gpurun_out/wtr_$C.log
done
cd $R
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/wtr_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("# 256 draws x 500 steps per launch; raw counter values (FETCH_SIZE / WRITE_SIZE in KiB per the guide's HBM section; gfx950: fetch x 2 for wide reads)")
for k, cs in acc.items():
    if "wgl" not in k: continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:12s} mean/launch = {sum(v)/len(v):.4g}  (n={len(v)})")
PY
find gpurun_out -name "*.csv" -size +1M -delete
