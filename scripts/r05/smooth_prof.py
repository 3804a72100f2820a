"""One smoother entry point at one shape for rocprofv3: usage smooth_prof.py headline|c3|h8 bf|rts"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
which = sys.argv[1] if len(sys.argv) > 1 else "headline"
kind = sys.argv[2] if len(sys.argv) > 2 else "bf"
mk, ft = {"headline": (lambda: W.structural(B=4096, n=1000), "standard"), "c3": (lambda: W.structural(B=8192, n=2000), "cholesky"),
          "h8": (lambda: W.structural(B=512, n=1000), "standard")}[which]
w = mk()
d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
res = eng.forward(*d, filter_type=ft, mvn_const="exact", store_ll_obs=False)
for _ in range(10):
    if kind == "bf":
        eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs)
    else:
        eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
torch.cuda.synchronize()
