"""GPU-box diagnostic: large-m forward path (MFMA GEMMs, workspace) vs the C oracle."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle import c_oracle as co  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402


def relerr(a, b):
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


eng = KalmanEngine()
cases = [
    ("dense_p1m48", W.random_dense(3, 12, 1, 48, 48, seed=1)),
    ("dense_p2m64", W.random_dense(3, 10, 2, 64, 8, seed=2)),
    ("dense_p3m70", W.random_dense(2, 8, 3, 70, 5, seed=3)),
    ("dense_p1m130", W.random_dense(2, 6, 1, 130, 3, seed=4)),
    ("c5_small", W.daily_seasonal(B=3, n=8)),
]
for name, w in cases:
    y = w["y"].copy()
    if w["n"] > 6:
        y[2] = np.nan
    args = (y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    for ft in ("standard", "cholesky", "univariate"):
        res = eng.forward(*args, filter_type=ft, mvn_const="exact")
        torch.cuda.synchronize()
        o = co.kalman_filter(*args, filter_type=ft, mvn_const="exact")
        print(f"{name:13s} m={w['m']:3d} p={w['p']} {ft:10s} kernel={eng.forward_kernel_name(w['B'], w['n'], w['p'], w['m'], w['r'])} "
              f"|dlogp| {np.max(np.abs(res.logp.cpu().numpy() - o['logp'])):.1e} "
              f"ll {relerr(res.ll_obs.cpu().numpy(), o['ll_obs']):.1e} "
              f"fa {relerr(res.filtered_states.cpu().numpy(), o['filtered_states']):.1e} "
              f"pa {relerr(res.predicted_states.cpu().numpy(), o['predicted_states']):.1e} "
              f"fP {relerr(res.filtered_covs.cpu().numpy(), o['filtered_covs']):.1e} "
              f"pP {relerr(res.predicted_covs.cpu().numpy(), o['predicted_covs']):.1e} "
              f"info {int(res.info.abs().sum())}", flush=True)
# C5-shaped timing on one GPU's share: B = 1024/8 = 128 draws, m = 367, T = 730 (logp only)
w = W.daily_seasonal(B=128, n=730)
d = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
kw = dict(store_ll_obs=True, store_filtered=False, store_predicted=False)
res = eng.forward(*d, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.forward(*d, out=res, **kw)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
flops = 128 * 730 * (4.0 * 367 ** 3)
print(f"c5 share B=128 n=730 m=367 logp-only: {dt:.3f} s -> {128 / dt:.1f} evals/s, "
      f"{flops / dt / 1e12:.2f} TFLOP/s on the two m^3 GEMMs")
