import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for m in (12, 16, 24, 28, 32, 40):
    w0 = W.random_dense(3, 20, 1, m, 1, seed=180 + m)
    P0 = w0["P0"].copy(); P0[:, 0, :] = 0.0; P0[:, :, 0] = 0.0
    flt0 = kn.kalman_filter(w0["y"], w0["a0"], P0, w0["T"], w0["Z"], w0["R"], w0["H"], w0["Q"])
    ra, rP = kn.kalman_smoother(w0["T"], w0["R"], w0["Q"], flt0["filtered_states"], flt0["filtered_covs"])
    sa, sP = eng.smooth(w0["T"], w0["R"], w0["Q"], flt0["filtered_states"], flt0["filtered_covs"])
    eng.set_forward_kernel("lds")
    la, lP = eng.smooth(w0["T"], w0["R"], w0["Q"], flt0["filtered_states"], flt0["filtered_covs"])
    eng.set_forward_kernel("auto")
    # condition of P_hat at a few steps
    Ph = np.einsum("bij,bjk,blk->bil", w0["T"], flt0["filtered_covs"][:, 5], w0["T"]) + np.einsum("bij,bjk,blk->bil", w0["R"], w0["Q"], w0["R"])
    ev = np.linalg.eigvalsh(Ph)
    print(f"m={m}: tile-vs-oracle {np.abs(sP.cpu().numpy()-rP).max():.3e}  lds-vs-oracle {np.abs(lP.cpu().numpy()-rP).max():.3e}  "
          f"|P_s| {np.abs(rP).max():.3e}  eig(P_hat) min/max {ev.min():.2e}/{ev.max():.2e}", flush=True)
