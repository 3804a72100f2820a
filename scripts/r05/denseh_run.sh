#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_denseh; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_r05.py -q -x -k "dense_H" > $O/pytest.txt 2>&1; tail -25 $O/pytest.txt
timeout 900 python3 -m pytest tests/test_gpu_r04.py tests/test_gpu_parity.py -q -x -k "dense or whiten or value_and_grad or c4" > $O/pytest2.txt 2>&1; tail -4 $O/pytest2.txt
for dh in "" "--dense-h"; do
  python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu $dh 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); o = j.get('other_entry_points', {})
        print('c4 $dh: forward', round(j['ms_per_step'], 2), 'ms; logp+grad', o.get('logp_and_gradient_one_call_ms'), 'ms; exact', o.get('logp_and_gradient_exact_recursion_ms'), o.get('error'))
"
done | tee $O/lines.txt
