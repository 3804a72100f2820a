"""A few forward passes of one workload through the split pipeline (for rocprofv3 --kernel-trace)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine

name = sys.argv[1] if len(sys.argv) > 1 else "headline"
full = (sys.argv[2] if len(sys.argv) > 2 else "full") == "full"
B = int(sys.argv[3]) if len(sys.argv) > 3 else None
eng = KalmanEngine()
w = W.WORKLOADS[name](**({"B": B} if B else {}))
dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
ft = "cholesky" if name == "c3" else "standard"
fkw = dict(filter_type=ft, mvn_const="exact", store_ll_obs=full, store_filtered=full, store_predicted=full)
if len(sys.argv) > 4:
    os.environ["KF_B4_SPLIT_DEBUG"] = sys.argv[4]
res = eng.forward(*dev, **fkw)
for _ in range(4):
    eng.forward(*dev, out=res, **fkw)
torch.cuda.synchronize()
