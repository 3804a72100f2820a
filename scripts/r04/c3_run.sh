#!/bin/bash
# C3 (8192 x 2000, the split pipeline): bench lines + the timeline of one pass (kernel trace)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r04_c3; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do python3 bench.py --steps 20 --warmup 5 --no-cpu --no-ops --workload c3 > $O/c3_$i.json 2> $O/c3_$i.err; done
python3 bench.py --steps 20 --warmup 5 --no-cpu --no-ops --workload c3 --exact-recursion > $O/c3_exact.json 2> $O/c3_exact.err
( cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $ROOT/scripts/r03/split_trace.py c3 full > $O/trace.log 2>&1 )
python3 scripts/r03/trace_timeline.py $O/trace > $O/c3_split_timeline.txt 2>&1
python3 - <<PY
import json
for f in ("c3_1","c3_2","c3_3","c3_exact"):
    j=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); r=j["roofline"]
    print(f, "ms/step %.4f kernel_ms %.4f frac %.3f sclk %.0f"%(j["ms_per_step"], r["kernel_ms"], r["frac"], r.get("sclk_mhz") or 0))
PY
grep -n "compact" $O/c3_split_timeline.txt | head -20
find $O -name "*kernel_trace.csv" -size +1M -delete
