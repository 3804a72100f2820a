#!/bin/bash
# regression sweep at the end of round 6: the fuzz scripts of every round against the current build (fixed tolerances inside each)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r06_fuzzall; mkdir -p $O
cat .kf_commit | tee $O/summary.txt
run() {  # name timeout script args...
  local name=$1 to=$2; shift 2
  echo "== $name: python3 $*" | tee -a $O/summary.txt
  timeout $to python3 "$@" > $O/$name.txt 2>&1; echo "   exit code $?" | tee -a $O/summary.txt
  grep -v -i "amdgpu.ids\|warn\|nanmean" $O/$name.txt | tail -3 | cut -c1-400 | tee -a $O/summary.txt
}
run gpu_fuzz 1200 scripts/gpu_fuzz.py 1500 7
run fuzz_r03 900 scripts/r03/fuzz_r03.py 400 9
run fuzz_r04 900 scripts/r04/fuzz_r04.py 600 11
run fuzz_r04b 900 scripts/r04/fuzz_r04b.py 600 13
run fuzz_r05 900 scripts/r05/fuzz_r05.py 400 17
run fuzz_smoothers 900 scripts/r06/fuzz_smoothers.py 600 19
run fuzz_chunked 900 scripts/r06/fuzz_chunked.py 400 23
run fuzz_tape_grad 900 scripts/r06/fuzz_tape_grad.py 300 29
