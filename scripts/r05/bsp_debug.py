import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
os.environ["KF_BSP_DEBUG"] = "1"
import workloads as W
from pymc_statespace_amd import KalmanEngine
from oracle import kalman_numpy as kn
eng = KalmanEngine()
for (B, n, s) in [(11, 40, 69), (3, 12, 366), (8, 10, 69), (11, 10, 100)]:
    w = W.daily_seasonal(B=B, n=n, s=s)
    args = (w["y"],) + tuple(w[k] for k in ("a0", "P0", "T", "Z", "R", "H", "Q"))
    res = eng.forward(*args, mvn_const="exact", store_filtered=False, store_predicted=True)
    torch.cuda.synchronize()
    orc = kn.kalman_filter(*args, mvn_const="exact")
    print(B, n, s, eng.last_forward_kernel(), np.abs(res.logp.cpu().numpy() - orc["logp"]).max(),
          np.abs(res.predicted_covs.cpu().numpy() - orc["predicted_covs"]).max(), flush=True)
