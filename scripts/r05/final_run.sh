#!/bin/bash
# round-5 evidence, all at the stamped commit (.kf_commit): the headline pin (bench lines, clocks, rocprofv3 kernel stats, HBM and SQ
# counters), kernel stats + HBM traffic of C2 / C3 / C4 / C5, bench lines of every BASELINE workload, the strong-scaling proxy table,
# the time-varying / dense-H lines, logp + gradient timings, the smoother from the filter's tapes (on chip and on the workspace),
# the sparse-transition step kernel's counters.   (the full GPU test suite + fuzz: scripts/r05/pytest_full.sh, its own lease)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
O=$ROOT/gpurun_out/r05_final; mkdir -p $O
scripts/r04/headline_pin.sh r05_pin_final > $O/headline_pin.txt 2>&1
cp gpurun_out/r05_pin_final/headline_summary.txt $O/headline_summary.txt
cp gpurun_out/r05_pin_final/clock_ramp.txt $O/clock_ramp.txt 2>/dev/null
cp gpurun_out/r05_pin_final/traffic_entry.json $O/traffic_headline.json 2>/dev/null
for w in c2 c3 c4 c5; do
  scripts/profile.sh r05_$w --workload $w > $O/${w}_summary.txt 2>&1
  cp gpurun_out/r05_$w/traffic_entry.json $O/traffic_$w.json 2>/dev/null
done
for w in headline c2 c3 c4 c5; do
  steps=20; [ $w = c5 ] && steps=3
  python3 bench.py --workload $w --steps $steps --warmup 5 > $O/bench_$w.json 2> $O/bench_$w.err
done
scripts/r04/strong_proxy.sh r05_strong_final > $O/strong_proxy.log 2>&1
cp gpurun_out/r05_strong_final/strong_proxy.md $O/strong_proxy.md
# time-varying and dense-H lines
for tv in Q R,Q T T,Q T,Z; do
  python3 bench.py --steps 10 --warmup 2 --no-cpu --no-ops --time-varying $tv 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('headline, time-varying $tv:', round(j['ms_per_step'], 3), 'ms', r.get('kernel'), 'frac', round(r['frac'], 3))
"
done > $O/tv_lines.txt 2>&1
for dh in "" "--dense-h"; do
  python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu $dh 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); o = j.get('other_entry_points', {})
        print('c4 $dh: forward', round(j['ms_per_step'], 2), 'ms; logp+grad one call', o.get('logp_and_gradient_one_call_ms'), 'ms; exact recursion', o.get('logp_and_gradient_exact_recursion_ms'), o.get('error'))
"
done > $O/denseh_lines.txt 2>&1
python3 scripts/r04/vg_time.py > $O/vg_time.txt 2>&1
python3 scripts/r05/smooth_bf_time.py 2>&1 | grep -v amdgpu.ids > $O/smooth_from_filter_time.txt
bash scripts/r05/smooth_prof.sh > $O/smooth_from_filter_prof.txt 2>&1
bash scripts/r05/bfbig_run.sh > $O/c5_smoother_from_filter.txt 2>&1
cp gpurun_out/r05_bfbig/kernels.txt $O/c5_smoother_from_filter_kernels.txt 2>/dev/null
BSP_PROF_ROWS=8 bash scripts/r05/bsp_prof.sh > $O/c5_sparse_step_counters.txt 2>&1
bash scripts/r05/c5_vg_prof.sh > $O/c5_value_and_grad_kernels.txt 2>&1
python3 scripts/r05/tv_grad_time.py 4096 2>&1 | grep -v amdgpu > $O/tv_gradient_smoother_time.txt
KF_BACKWARD_TV_MFMA=0 KF_SMOOTHER_TV_MFMA=0 python3 scripts/r05/tv_grad_time.py 4096 2>&1 | grep -v amdgpu | sed 's/^/KF_BACKWARD_TV_MFMA=0 KF_SMOOTHER_TV_MFMA=0  /' >> $O/tv_gradient_smoother_time.txt
python3 scripts/r05/denseh_time.py 2>&1 | grep -v amdgpu > $O/denseh_same_data.txt
bash scripts/r05/denseh_prof.sh >> $O/denseh_same_data.txt 2>&1
cat .kf_commit
find gpurun_out -name "*.csv" -size +1M -delete
