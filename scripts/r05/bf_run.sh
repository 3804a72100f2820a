#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_bf; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_r05.py -x -q > $O/pytest.txt 2>&1; tail -15 $O/pytest.txt
timeout 600 python3 scripts/r05/smooth_bf_time.py 2>&1 | grep -v amdgpu.ids | tee $O/time.txt
