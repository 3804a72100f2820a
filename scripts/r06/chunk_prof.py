"""One shape of the chunked pipeline, for rocprofv3 --kernel-trace --stats: python scripts/r06/chunk_prof.py [B] [T] [full|logp]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
full = (sys.argv[3] if len(sys.argv) > 3 else "full") == "full"
eng = KalmanEngine()
w = W.structural(B=B, n=T)
dv = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
for _ in range(30):
    eng.forward(*dv, mvn_const="exact", store_ll_obs=full, store_filtered=full, store_predicted=full)
torch.cuda.synchronize()
