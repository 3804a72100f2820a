#!/bin/bash
# fast iteration on the forward block kernel: headline bench (full tapes, logp only), C3, a parity subset
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; O=gpurun_out/quick; mkdir -p $O
for i in 1 2; do python3 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu --no-ops > $O/h$i.json 2>$O/h$i.err; done
python3 bench.py --gpus 1 --steps 40 --warmup 5 --no-cpu --no-ops --logp-only > $O/hl.json 2>$O/hl.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-ops --workload c3 > $O/c3.json 2>$O/c3.err
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu --no-ops --workload headline --batch 512 > $O/h512.json 2>$O/h512.err
python3 - <<PY
import json
for f in ("h1","h2","hl","c3","h512"):
    try:
        j=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1]); r=j["roofline"]
        print(f, "ms/step %.4f kernel_ms %.4f frac %.3f sclk %.0f Mcycles %.3f"%(j["ms_per_step"], r["kernel_ms"], r["frac"], r.get("sclk_mhz") or 0, r.get("kernel_mcycles_at_sclk") or 0))
    except Exception as e: print(f, "ERR", e, open("$O/%s.err"%f).read()[-300:])
PY
python3 -m pytest tests/test_gpu_parity.py -x -q -k "b4 or structural or block or steady" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
