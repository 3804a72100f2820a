#!/bin/bash
# C5 share: A/B of the fused forward against KF_BIG_FUSE=0, the C5 tests, per-kernel averages of the A/B under rocprofv3
python scripts/r03/c5_ab.py 2>&1 | tail -6
python -m pytest tests/test_gpu_c5.py -q 2>&1 | tail -4
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/c5prof
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5prof -o c5 -- python3 $R/scripts/r03/c5_ab.py > /dev/null 2>&1
cd $R && python - <<'PY'
import sqlite3
db = sqlite3.connect('gpurun_out/c5prof/c5_results.db')
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
for r in db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3 from {kd} d join {ks} s on d.kernel_id=s.id group by 1"):
    if 'big_' in r[0]: print(r[0][14:70], r[1], round(r[2], 1))
PY
