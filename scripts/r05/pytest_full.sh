#!/bin/bash
# the full GPU suite at the stamped commit -> gpurun_out/r05_full/pytest_gpu.txt, then the random-shape sweep
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_full; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt; cat .kf_commit
for s in 61 62 63; do timeout 1500 python3 scripts/r05/fuzz_r05.py 400 $s 2>&1 | grep -v amdgpu.ids; done | tee $O/fuzz_summary.txt
