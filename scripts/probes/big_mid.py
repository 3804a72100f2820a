"""One forward / gradient / smoother pass at mid-size m on the HBM-workspace path (for rocprofv3)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
B, n, m = int(os.environ.get("NB_B", 4096)), int(os.environ.get("NB_N", 50)), int(os.environ.get("NB_M", 64))
w = W.random_dense(64, n, 1, m, max(1, m // 3), seed=7)
rep = B // 64
dev = [eng.to_device(w["y"])] + [eng.to_device(np.tile(w[k], (rep,) + (1,) * (w[k].ndim - 1))) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
what = os.environ.get("NB_WHAT", "fwd")
res = eng.forward(*dev); torch.cuda.synchronize()
if "smooth" in what:
    eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs)
if "grad" in what:
    g = torch.ones(B, dtype=torch.float64, device="cuda")
    eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=g)
torch.cuda.synchronize()
print("done", eng.forward_kernel_name(B, n, 1, m, max(1, m // 3)))
