#!/bin/bash
# sparse-transition forward (bsp_step_kernel): parity tests, C5 timing against the MFMA products, kernel split
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_bsp; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_r05.py -q -x -k "sparse_transition or zero_blocks" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 900 python3 -m pytest tests/test_abi.py tests/test_gpu_c5.py -q -x > $O/pytest_c5.txt 2>&1; tail -3 $O/pytest_c5.txt
for cfg in "1 16" "1 8" "0 8"; do
  set -- $cfg
  KF_EXPERIMENTS=1 KF_BIG_SPARSE=$1 KF_BSP_ROWS=$2 python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu --no-ops 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('KF_BIG_SPARSE=$1 rows=$2 c5:', round(j['ms_per_step'], 2), 'ms per 128 draws', j['value'], j['unit'])
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
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 /tmp/c5_small.py > $O/small.txt 2>&1
grep -v "rocprofv3\|^W2\|^E2\|amdgpu.ids" $O/small.txt | tail -2
python3 $ROOT/scripts/kernel_stats.py $O/kt 8 | tee $O/kernels.txt
