#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03_final
python3 bench.py --workload c5 --steps 2 --warmup 2 > gpurun_out/r03_final/bench_c5.json 2> gpurun_out/r03_final/bench_c5.err
python3 scripts/r03/fuzz_r03.py 400 23 > gpurun_out/r03_final/fuzz.txt 2>&1
tail -3 gpurun_out/r03_final/fuzz.txt
python3 -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r03_final/pytest_gpu.txt
python3 bench.py > gpurun_out/r03_final/bench_default.json 2> gpurun_out/r03_final/bench_default.err
tail -c 400 gpurun_out/r03_final/bench_default.json
