#!/bin/bash
# numbers for the time-varying and the dense-H paths at the headline / C4 shapes
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=$ROOT/gpurun_out/r04_tv; mkdir -p $O
python3 bench.py --steps 5 --warmup 2 --workload headline --time-varying T --no-cpu > $O/headline_tvT.json 2> $O/headline_tvT.err
python3 bench.py --steps 5 --warmup 2 --workload headline --time-varying T,Z --no-cpu > $O/headline_tvTZ.json 2> $O/headline_tvTZ.err
python3 bench.py --steps 5 --warmup 2 --workload headline --time-varying Q --no-cpu > $O/headline_tvQ.json 2> $O/headline_tvQ.err
python3 bench.py --steps 3 --warmup 1 --workload c4 --dense-h --no-cpu > $O/c4_denseH.json 2> $O/c4_denseH.err
python3 bench.py --steps 3 --warmup 1 --workload c4 --dense-h --batch 2048 > $O/c4_denseH_b2048.json 2> $O/c4_denseH_b2048.err
python3 bench.py --steps 3 --warmup 1 --workload c4_m10 --dense-h --no-cpu > $O/c4m10_denseH.json 2> $O/c4m10_denseH.err
python3 - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/*.json")):
    try:
        j=json.loads(open(f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(os.path.basename(f), "ms/step %.3f kernel %s frac %.3f"%(j["ms_per_step"], r["kernel"], r["frac"]), {k:round(v,3) for k,v in j.get("other_entry_points",{}).items() if isinstance(v,float)})
    except Exception as e: print(f, "ERR", e, open(f.replace(".json",".err")).read()[-400:])
PY
