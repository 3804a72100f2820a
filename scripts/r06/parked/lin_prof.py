"""Headline forward, all tapes, ten calls with the linear covariance-tape staging and ten with the rings (for rocprofv3: the two are
different instantiations of kf_fwd_b4_kernel, <3, true, 15, 4> and <3, true, 15, 0>)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

eng = KalmanEngine()
w = W.structural(B=4096, n=1000)
dv = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
for lin in ("1", "0"):
    os.environ["KF_TAPE_LINEAR"] = lin
    for _ in range(10):
        eng.forward(*dv, mvn_const="exact")
    torch.cuda.synchronize()
