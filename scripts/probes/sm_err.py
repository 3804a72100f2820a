import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for (n, m) in [(63, 21), (64, 21), (129, 37), (63, 50)]:
    B, p, r = 3, 1, m
    w = W.random_dense(B, n, p, m, r, seed=1900 + n + m)
    rng = np.random.default_rng(n + m)
    yb = np.repeat(w["y"][None], B, axis=0) + 0.0
    yb = yb + 0.1 * rng.standard_normal(yb.shape)
    yb[:, n // 2] = np.nan
    ref0 = kn.kalman_filter(yb, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    ra, rP = kn.kalman_smoother(w["T"], w["R"], w["Q"], ref0["filtered_states"], ref0["filtered_covs"])
    sa, sP = eng.smooth(w["T"], w["R"], w["Q"], ref0["filtered_states"], ref0["filtered_covs"])
    eng.set_forward_kernel("lds"); la, lP = eng.smooth(w["T"], w["R"], w["Q"], ref0["filtered_states"], ref0["filtered_covs"]); eng.set_forward_kernel("auto")
    e1 = np.abs(sP.cpu().numpy() - rP).reshape(B, n, -1).max(axis=2).max(axis=0)
    e2 = np.abs(lP.cpu().numpy() - rP).reshape(B, n, -1).max(axis=2).max(axis=0)
    asym = np.abs(ref0["filtered_covs"] - np.swapaxes(ref0["filtered_covs"], -1, -2)).max()
    print(n, m, "tile err by t (every 8):", np.array2string(e1[::8], precision=1), "\n      lds err:", np.array2string(e2[::8], precision=1),
          "rho(T)", np.abs(np.linalg.eigvals(w["T"])).max().round(3), "Pf asym", asym)
