#!/bin/bash
# the headline roofline pinned in ONE lease -- un-profiled bench (the driver's command line), the same
# from a cold clock, a sustained run, rocprofv3 --kernel-trace --stats, the --pmc passes -- every leg bracketed by a shader / memory
# clock and power reading (amdgpu hwmon + rocm-smi, read by this wrapper, never under rocprofv3).
#   usage: scripts/r04/headline_pin.sh [tag]      (outputs under gpurun_out/<tag>/, summary in <tag>/pin_summary.txt)
set -u
TAG=${1:-r04_pin}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
clk() {  # one line: label, sclk MHz, mclk MHz, socket W, junction C
  local f p m t
  f=$(cat $HW/freq1_input 2>/dev/null); m=$(cat $HW/freq2_input 2>/dev/null); p=$(cat $HW/power1_input 2>/dev/null); t=$(cat $HW/temp2_input 2>/dev/null)
  echo "[clk] $1: sclk $((f / 1000000)) MHz  mclk $((m / 1000000)) MHz  power $((p / 1000000)) W  temp $((t / 1000)) C  ($(date +%T.%N))"
}
poll() {  # background sampler while a leg runs: 50 ms
  while true; do clk "$1"; sleep 0.05; done
}
leg() {  # leg <name> <command...>: clocks before, sampled during (max / mean reported), after
  local name=$1; shift
  clk "$name before" >> $O/clocks.txt
  poll "$name during" > $O/poll_$name.txt 2>/dev/null &
  local P=$!
  "$@"
  kill $P 2>/dev/null; wait $P 2>/dev/null
  awk -v n="$name" '{s+=$5; if($5>mx)mx=$5; pw+=$11; if($11>pmx)pmx=$11; c++} END{if(c)printf "[clk] %s during: %d samples, sclk mean %.0f max %.0f MHz, power mean %.0f max %.0f W (incl. process start-up at idle clocks)\n", n, c, s/c, mx, pw/c, pmx}' $O/poll_$name.txt >> $O/clocks.txt
  clk "$name after" >> $O/clocks.txt
}
: > $O/clocks.txt
DRV="--gpus 1 --steps 20 --warmup 5"
leg driver_cmd      bash -c "python3 bench.py $DRV > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err"
leg cold_no_spinup  bash -c "python3 bench.py $DRV --no-spinup --no-cpu --no-ops > $O/bench_cold.json 2> $O/bench_cold.err"
leg sustained       bash -c "python3 bench.py --steps 2000 --warmup 20 --no-cpu --no-ops > $O/bench_sustained.json 2> $O/bench_sustained.err"
leg exact           bash -c "python3 bench.py $DRV --exact-recursion --no-cpu --no-ops > $O/bench_exact.json 2> $O/bench_exact.err"
python3 scripts/r04/clock_ramp.py > $O/clock_ramp.txt 2>&1
cd /tmp
B="$ROOT/bench.py $DRV --no-cpu --no-ops"
leg rocprof_kt  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $B > $O/bench_kt.json 2> $O/kt.log
leg pmc_fetch   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $B > $O/bench_fetch.json 2> $O/fetch.log
leg pmc_write   rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $B > $O/bench_write.json 2> $O/write.log
leg pmc_sq1     rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $O/pmc_sq1 -- python3 $B > $O/bench_sq1.json 2> $O/sq1.log
leg pmc_sq2     rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES --output-format csv -d $O/pmc_sq2 -- python3 $B > $O/bench_sq2.json 2> $O/sq2.log
leg pmc_sq3     rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_sq3 -- python3 $B > $O/bench_sq3.json 2> $O/sq3.log
cd "$ROOT"
python3 scripts/summarize_profile.py $O > $O/headline_summary.txt 2>&1
python3 scripts/r04/pin_summary.py $O > $O/pin_summary.txt 2>&1
cat $O/pin_summary.txt
find $O -name "*.csv" -size +2M -delete
