#!/bin/bash
# the time-parallel steady-state adjoint + placement hint: parity, logp + gradient times, kernel stats, bench line
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r04_scan; mkdir -p $O
python3 -m pytest tests/test_gpu_r04.py tests/test_gpu_switches.py tests/test_pytensor_op.py -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err; python3 -c "
import json; j=json.loads(open('$O/bench_headline.json').read().strip().splitlines()[-1]); print(j['ms_per_step'], j['roofline']['frac'], j['other_entry_points'])"
python3 scripts/r04/vg_time.py hint > $O/vg_hint.txt 2>&1; cat $O/vg_hint.txt
