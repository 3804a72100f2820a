#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_c5grad; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_r05.py -q -x -k "zero_blocks" 2>&1 | tail -3
timeout 1200 python3 -m pytest tests/test_gpu_c5.py -q -x 2>&1 | tail -3
python3 bench.py --workload c5 --steps 2 --warmup 1 --no-cpu 2>$O/bench_err.txt | tee $O/bench_c5.json | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l)
        print(json.dumps(j.get('other_entry_points'), indent=1))
" | tee $O/c5_ops.txt
