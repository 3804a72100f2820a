#!/bin/bash
# workspace-path from-filter smoother: parity tests, C5 per-step timing next to the RTS smoother, kernel split
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_bfbig; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_r05.py -q -x -k "workspace_path or rejects or golden or restatement" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 600 python3 -m pytest tests/test_abi.py tests/test_gpu_c5.py -q -x > $O/pytest_c5.txt 2>&1; tail -3 $O/pytest_c5.txt
python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu 2>$O/bench_err.txt | tee $O/bench_c5.json | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print(json.dumps(j.get('other_entry_points'), indent=1))
" | tee $O/c5_ops.txt
tail -3 $O/bench_err.txt
cat > /tmp/bfbig_small.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.daily_seasonal(B=128, n=12)
d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
res = eng.forward(*d, mvn_const="exact", store_ll_obs=False, store_filtered=True, store_predicted=True)
for _ in range(2):
    sa, sP = eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs)
    sa2, sP2 = eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
torch.cuda.synchronize()
print(float((sa - sa2).abs().max()), float((sP - sP2).abs().max()), float(sP2.abs().max()))
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 /tmp/bfbig_small.py > $O/small.txt 2>&1
tail -2 $O/small.txt
python3 $ROOT/scripts/kernel_stats.py $O/kt 16 | tee $O/kernels.txt
