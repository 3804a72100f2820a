#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_c5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_r05.py -q -x -k "big_path" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_c5.py -q -x > $O/pytest_c5.txt 2>&1; tail -3 $O/pytest_c5.txt
for tm in 1 0; do
  KF_BIG_TMASK=$tm python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu --no-ops 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('KF_BIG_TMASK=$tm c5:', round(j['ms_per_step'], 2), 'ms per 128 draws', 'frac', round(r['frac'], 3), 'issued TF', round(r['achieved'], 1), 'ref-sequence TF', round(r.get('reference_flops_tflops', 0), 1))
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
print(float(res.logp[0]))
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 /tmp/c5_small.py > /dev/null 2>&1
python3 $ROOT/scripts/kernel_stats.py $O/kt 6
