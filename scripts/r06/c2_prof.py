"""C2's forward (kf_fwd_lane_kernel) alone, for rocprofv3 / SQ counters: python scripts/r06/c2_prof.py [exact]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

eng = KalmanEngine()
w = W.arma11(B=4096, n=1000)
dv = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
ex = True if (len(sys.argv) > 1 and sys.argv[1] == "exact") else None
for _ in range(30):
    eng.forward(*dv, mvn_const="exact", exact_recursion=ex)
torch.cuda.synchronize()
