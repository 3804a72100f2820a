#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_tvq; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_r05.py tests/test_gpu_r04.py tests/test_pytensor_op.py -q -x -k "time_varying or smoother_nodes or ll_obs" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
for tv in Q R,Q T T,Q; do
  python3 bench.py --steps 10 --warmup 2 --no-cpu --no-ops --time-varying $tv 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('time-varying $tv:', j['ms_per_step'], 'ms', r.get('kernel'), 'frac', round(r['frac'], 3), 'bytes', r.get('algorithmic_bytes_per_launch'))
"
done | tee $O/lines.txt
