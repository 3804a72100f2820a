#!/bin/bash
# AUTO against every forced kernel family over grids of shapes (forward p = 1 and p > 1, logp + gradient, smoother) -> gpurun_out/r04_dispatch/
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r04_dispatch; mkdir -p $O
timeout 900 python3 scripts/r04/fwd_dispatch_sweep.py 2>&1 | grep -v "amdgpu\|Warning\|return (y" > $O/fwd_dispatch_sweep.txt
timeout 900 python3 scripts/r04/fwd_mp_dispatch_sweep.py 2>&1 | grep -v "amdgpu\|Warning\|return (y" > $O/fwd_mp_dispatch_sweep.txt
timeout 900 python3 scripts/r04/vg_dispatch_sweep.py 2>&1 | grep -v "amdgpu\|Warning\|return (y" > $O/vg_dispatch_sweep.txt
timeout 900 python3 scripts/r04/vg_rowshapes.py 2>&1 | grep -v "amdgpu\|Warning\|return (y" > $O/vg_rowshapes.txt
timeout 900 python3 scripts/r04/smooth_dispatch_sweep.py 2>&1 | grep -v "amdgpu\|Warning\|return (y" > $O/smooth_dispatch_sweep.txt
grep -c . $O/*.txt; grep -h "<--" $O/*.txt | cut -c1-200 | head -40
cat .kf_commit
