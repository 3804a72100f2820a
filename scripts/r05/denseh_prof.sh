#!/bin/bash
# C4 with a dense H: kernel split of kf_filter_value_and_grad
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_denseh_prof; mkdir -p $O
cat > /tmp/dh_vg.py <<'PY'
import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.varmax()
rng = np.random.default_rng(0)
p = w["p"]
A = 0.1 * rng.standard_normal((w["B"], p, p))
H = w["H"] + A @ np.swapaxes(A, 1, 2)
d = [eng.to_device(w["y"])] + [eng.to_device(x) for x in (w["a0"], w["P0"], w["T"], w["Z"], w["R"], H, w["Q"])]
for _ in range(3):
    out = eng.value_and_grad(*d, mvn_const="exact")
torch.cuda.synchronize()
print(float(out[0][0]))
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 /tmp/dh_vg.py > $O/small.txt 2>&1
python3 $ROOT/scripts/kernel_stats.py $O/kt 20 | tee $O/kernels.txt
