#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_bsptime; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_r05.py -q -x -k "sparse_transition or zero_blocks" 2>&1 | tail -2
timeout 600 python3 -m pytest tests/test_gpu_c5.py -q -x -k "fused" 2>&1 | tail -2
for sp in 1 0; do
KF_BIG_SPARSE=$sp python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu --no-ops 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('KF_BIG_SPARSE=$sp c5:', round(j['ms_per_step'], 2), 'ms per 128 draws', round(j['value'], 1), j['unit'])
"
done | tee $O/lines.txt
