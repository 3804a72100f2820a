#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_tvsm; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_parity.py tests/test_op_wiring.py tests/test_pytensor_op.py -q -x -k "time_varying" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
python3 scripts/r05/tv_grad_time.py 4096 2>&1 | grep -v amdgpu | tee $O/tv_time.txt
