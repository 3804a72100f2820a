import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
m = 52
w0 = W.random_dense(3, 12, 1, m, m, seed=280 + m)
Q = w0["Q"].copy(); Q[1] = 0.0
P0 = w0["P0"].copy(); P0[1] = 0.0
flt0 = kn.kalman_filter(w0["y"], w0["a0"], P0, w0["T"], w0["Z"], w0["R"], w0["H"], Q)
sa0, sP0 = eng.smooth(w0["T"], w0["R"], Q, flt0["filtered_states"], flt0["filtered_covs"])
sa0, sP0 = sa0.cpu().numpy(), sP0.cpu().numpy()
for d in range(3):
    print("draw", d, "nan rows a:", np.isnan(sa0[d]).all(axis=1).astype(int), "any:", np.isnan(sa0[d]).any(axis=1).astype(int))
    print("   nan P per t:", np.isnan(sP0[d]).reshape(12, -1).mean(axis=1).round(2))
print("filtered covs draw1 max", np.abs(flt0["filtered_covs"][1]).max())
