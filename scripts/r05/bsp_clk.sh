#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
timeout 600 python3 -m pytest tests/test_gpu_r05.py -q -x -k "sparse_transition" 2>&1 | tail -2
for cfg in ${BSP_CFGS:-2:0 2:1 2:3 2:6 2:12 1:0 1:6 4:0 4:2}; do
  rows=${cfg%%:*}; nbw=${cfg##*:}
  if [ "$nbw" = "0" ]; then unset KF_BSP_NBW; else export KF_BSP_NBW=$nbw; fi
  KF_BSP_DEBUG=1 KF_BSP_ROWS=$rows python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu --no-ops 2>/tmp/err.txt | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print('rows=$rows nbw=$nbw c5:', round(j['ms_per_step'], 2), 'ms per 128 draws', round(j['value'], 1), j['unit'])
"
  grep clocks /tmp/err.txt | tail -1
done
