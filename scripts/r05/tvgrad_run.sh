#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_tvgrad; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_r05.py tests/test_gpu_parity.py -q -x -k "time_varying_gradient" > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
python3 scripts/r05/tv_grad_time.py 4096 2>&1 | grep -v amdgpu | tee $O/tv_grad_time.txt
KF_BACKWARD_TV_MFMA=0 python3 scripts/r05/tv_grad_time.py 4096 2>&1 | grep -v amdgpu | sed 's/^/KF_BACKWARD_TV_MFMA=0  /' | tee -a $O/tv_grad_time.txt
