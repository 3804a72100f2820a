#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_fuzz; mkdir -p $O
timeout 600 python3 -m pytest tests/test_gpu_r05.py tests/test_pytensor_op.py -q -x -k "ill_conditioned or smoother_nodes" 2>&1 | tail -4
for s in 51 52 53; do timeout 1500 python3 scripts/r05/fuzz_r05.py 400 $s 2>&1 | grep -v amdgpu.ids; done | tee $O/fuzz_summary.txt
