"""The RTS smoother alone at the headline shape (for rocprofv3 passes): KF_SMOOTHER_STEADY from the environment."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "headline"]()
dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
res = eng.forward(*dev, store_ll_obs=False, store_predicted=False)
for _ in range(3):
    out = eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs)
torch.cuda.synchronize()
