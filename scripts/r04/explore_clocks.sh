#!/bin/bash
# what clock / power telemetry an ordinary user can read on the GPU box
O=gpurun_out/r04_explore; mkdir -p $O
{
echo "== which"; which rocm-smi amd-smi rocminfo
echo "== rocm-smi --showclocks --showpower"; rocm-smi --showclocks --showpower 2>&1 | head -40
echo "== rocm-smi json"; rocm-smi --showclocks --showpower --showtemp --json 2>&1 | head -c 3000
echo; echo "== amd-smi metric -c -p"; timeout 30 amd-smi metric -c -p 2>&1 | head -80
echo "== amd-smi static limits"; timeout 30 amd-smi static -l 2>&1 | head -40
echo "== sysfs"
for c in /sys/class/drm/card*/device; do
  echo "-- $c"; ls $c 2>/dev/null | tr '\n' ' '; echo
  for f in pp_dpm_sclk pp_dpm_mclk pp_dpm_fclk power_dpm_force_performance_level; do echo "[$f]"; cat $c/$f 2>&1 | head -20; done
  for h in $c/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in freq1_input freq2_input power1_average power1_input power1_cap temp1_input; do echo -n "[$f] "; cat $h/$f 2>&1; done; done
done
} > $O/telemetry.txt 2>&1
# clocks sampled while the headline bench runs (2000 steps ~ 3.5 s of kernel time)
( for i in $(seq 1 60); do date +%s.%N; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|Power|fclk" ; sleep 0.25; done ) > $O/smi_poll.txt 2>&1 &
POLL=$!
python3 bench.py --steps 2000 --warmup 20 --no-cpu --no-ops > $O/bench_long.json 2> $O/bench_long.err
kill $POLL 2>/dev/null
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
tail -c 600 $O/bench_long.json
