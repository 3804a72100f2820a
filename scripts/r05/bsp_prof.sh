#!/bin/bash
# C5 forward on the sparse-transition path against the MFMA products; kernel stats and SQ / TCC counters of bsp_step_kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_bspprof; mkdir -p $O
for sp in 1 0; do
  KF_BIG_SPARSE=$sp python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu --no-ops 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('KF_BIG_SPARSE=$sp c5:', round(j['ms_per_step'], 2), 'ms per 128 draws', round(j['value'], 1), j['unit'], j['roofline']['kernel'])
"
done | tee $O/lines.txt
cat > /tmp/c5_small.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.daily_seasonal(B=128, n=48)
d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
for _ in range(3):
    res = eng.forward(*d, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
torch.cuda.synchronize()
print(float(res.logp[0]), eng.last_forward_kernel())
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 /tmp/c5_small.py > /dev/null 2>&1
python3 $ROOT/scripts/kernel_stats.py $O/kt 4
for c in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INST_LEVEL_VMEM" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum SQ_WAIT_ANY" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  tag=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$tag -- python3 /tmp/c5_small.py > /dev/null 2> $O/pmc_$tag.log
  python3 - $O/pmc_$tag <<'PY'
import csv, glob, sys, os
from collections import defaultdict
d = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "bsp_step" not in k: continue
    for c, v in sorted(cs.items()):
        print(f"   {k[:24]:24s} {c:28s} mean/launch = {sum(v)/len(v):.6g}  (n={len(v)})")
PY
done 2>&1 | tee $O/pmc.txt
find $O -name "*.csv" -size +1M -delete
