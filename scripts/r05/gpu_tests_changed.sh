#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_tests; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_r05.py tests/test_pytensor_op.py tests/test_gpu_switches.py tests/test_gpu_distributed.py -q > $O/pytest_changed.txt 2>&1; tail -8 $O/pytest_changed.txt
timeout 300 python3 scripts/r05/smooth_bf_time.py 2>&1 | grep -v amdgpu.ids | tee $O/time.txt
