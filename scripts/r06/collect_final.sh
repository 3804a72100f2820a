#!/bin/bash
# copy the evidence of scripts/r06/final_run.sh (gpurun_out/r06_final) and of the full GPU suite (gpurun_out/r06_full) into profiles/r06/
# under fixed names; COMMIT.txt = the commit the leases ran at
cd "$(dirname "$0")/../.."; F=gpurun_out/r06_final; P=profiles/r06
for f in headline_pin headline_summary headline_b512_summary c2_summary smooth_lane_time chunk_check; do
  [ -f $F/$f.txt ] && cp $F/$f.txt $P/$f.txt
done
[ -f $F/strong_proxy.md ] && cp $F/strong_proxy.md $P/strong_proxy.md
for w in headline c2 c3 c4 headline_b512 headline_b256; do [ -s $F/bench_$w.json ] && cp $F/bench_$w.json $P/bench_$w.json; done
python3 scripts/merge_traffic.py $F/traffic_*.json > /dev/null
C=$(cat $F/COMMIT.txt)
[ -f gpurun_out/r06_full/pytest_gpu.txt ] && cp gpurun_out/r06_full/pytest_gpu.txt $P/pytest_gpu_$C.txt
echo $C > $P/COMMIT.txt
cat $P/COMMIT.txt; tail -2 $P/pytest_gpu_$C.txt
