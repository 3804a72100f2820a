#!/bin/bash
# the time-parallel steady-state adjoint: parity, logp + gradient times, kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r04_scan; mkdir -p $O
python3 -m pytest tests/test_gpu_r04.py -x -q -k "steady or adjoint or value_and_grad" > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
python3 scripts/r04/vg_time.py > $O/vg_time.txt 2>&1; cat $O/vg_time.txt
scripts/r04/vg_prof.sh > $O/vg_kernel_stats.txt 2>&1; cat $O/vg_kernel_stats.txt
