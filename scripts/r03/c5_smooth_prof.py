"""C5 share smoother (kf_smoother_big.hip), 128 draws x 60 steps: for rocprofv3 --kernel-trace --stats."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
w = W.daily_seasonal(B=128, n=60)
dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
res = eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_predicted=False)
sa, sP = eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs); torch.cuda.synchronize()
t0 = time.perf_counter(); sa, sP = eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs); torch.cuda.synchronize()
print(f"smoother: {(time.perf_counter() - t0) / 59 * 1e3:.3f} ms per step")
