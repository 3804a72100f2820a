#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r04_smooth; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_r04.py -x -q -k "smoother" > $O/pytest.txt 2>&1; tail -30 $O/pytest.txt
