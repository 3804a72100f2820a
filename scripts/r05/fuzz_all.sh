#!/bin/bash
# regression sweep at the end of the round: the fuzz scripts of every earlier round against the current build
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r05_fuzzall; mkdir -p $O
cat .kf_commit
echo "== gpu_fuzz"; timeout 1200 python3 scripts/gpu_fuzz.py 1500 7 > $O/gpu_fuzz.txt 2>&1; grep -v amdgpu.ids $O/gpu_fuzz.txt | tail -4 | cut -c1-600
echo "== fuzz_r03"; timeout 900 python3 scripts/r03/fuzz_r03.py 400 9 > $O/fuzz_r03.txt 2>&1; grep -v amdgpu.ids $O/fuzz_r03.txt | tail -4 | cut -c1-600
echo "== fuzz_r04"; timeout 900 python3 scripts/r04/fuzz_r04.py 600 11 > $O/fuzz_r04.txt 2>&1; grep -v amdgpu.ids $O/fuzz_r04.txt | tail -4 | cut -c1-600
echo "== fuzz_r04b"; timeout 900 python3 scripts/r04/fuzz_r04b.py 600 13 > $O/fuzz_r04b.txt 2>&1; grep -v amdgpu.ids $O/fuzz_r04b.txt | tail -4 | cut -c1-600
