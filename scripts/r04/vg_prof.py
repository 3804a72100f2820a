"""One shape of kf_filter_value_and_grad for rocprofv3 --kernel-trace --stats: usage vg_prof.py headline|h8|c3 [exact|steady]"""
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
mode = {"exact": True, "steady": False, "steady_nan": False}.get(sys.argv[2] if len(sys.argv) > 2 else "", None)
w, ft = {"headline": (lambda: W.structural(B=4096, n=1000), "standard"), "c3": (lambda: W.structural(B=8192, n=2000), "cholesky"),
         "h8": (lambda: W.structural(B=512, n=1000), "standard")}[which]
w = w()
if len(sys.argv) > 2 and sys.argv[2] == "steady_nan":   # a missing last observation: no convergence test ever runs, every draw takes all steps
    w["y"] = w["y"].copy()
    w["y"][-1] = float("nan")
d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
for _ in range(40):
    eng.value_and_grad(*d, filter_type=ft, mvn_const="exact", exact_recursion=mode)
torch.cuda.synchronize()
