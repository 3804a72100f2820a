#!/bin/bash
# the full GPU suite at the stamped commit -> gpurun_out/r05_full/pytest_gpu.txt
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_full; mkdir -p $O
python3 -m pytest tests -m gpu -q > $O/pytest_gpu.txt 2>&1
tail -5 $O/pytest_gpu.txt; cat .kf_commit
bash scripts/r05/tvq_run.sh | tail -5
