#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_c5; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_r05.py tests/test_gpu_parity.py -q -x -k "big_path or plain_cpp" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
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
bash scripts/r05/c5_gemm_prof.sh > $O/gemm_prof.txt 2>&1; tail -45 $O/gemm_prof.txt
