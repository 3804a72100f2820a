#!/bin/bash
# C5 value-and-gradient (128 draws x 24 steps): per-kernel time
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_c5vg; mkdir -p $O
cat > /tmp/c5_vg.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.daily_seasonal(B=128, n=24)
d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
for _ in range(2):
    out = eng.value_and_grad(*d, mvn_const="exact")
torch.cuda.synchronize()
print(float(out[0][0]))
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 /tmp/c5_vg.py > $O/small.txt 2>&1
python3 $ROOT/scripts/kernel_stats.py $O/kt 20 | tee $O/kernels.txt
