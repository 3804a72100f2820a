import sys
import numpy as np
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
eng.set_forward_kernel("mfma")
for m in (8, 10, 13):
    w = W.random_dense(3, 20, 1, m, max(1, m // 2), seed=500 + 17 + m)
    res = eng.forward(w["y"], w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    for name in ("filtered_covs", "predicted_covs"):
        x = getattr(res, name).cpu().numpy()
        d = x - np.swapaxes(x, -1, -2)
        idx = np.argwhere(d != 0)
        print(m, name, "n_asym", len(idx), "max", np.abs(d).max(), "first", idx[:6].tolist())
