"""C2 value_and_grad (4096 x 1000, ARMA(1,1)), ten calls -- for rocprofv3 kernel traces (scripts/r06/kt.sh)."""
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
for _ in range(10):
    eng.value_and_grad(*dv, mvn_const="exact")
torch.cuda.synchronize()
