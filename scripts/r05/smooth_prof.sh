#!/bin/bash
# kf_smoother_from_filter at the headline / C3 shapes: kernel stats, HBM traffic and SQ counters (separate --pmc passes)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_smooth_prof; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for w in headline c3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$w -- python3 $ROOT/scripts/r05/smooth_prof.py $w bf > /dev/null 2>&1
  echo "== $w: rocprofv3 --kernel-trace --stats"; python3 $ROOT/scripts/kernel_stats.py $O/kt_$w 4
done
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -- python3 $ROOT/scripts/r05/smooth_prof.py headline bf > /dev/null 2> $O/pmc_$tag.log
  python3 - $O/pmc_$tag <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "kf_smooth_bf" not in k: continue
    for c, v in sorted(cs.items()):
        print(f"   {k[:40]:40s} {c:28s} mean/launch = {sum(v)/len(v):.6g}  (n={len(v)})")
PY
done
find $O -name "*.csv" -size +1M -delete
