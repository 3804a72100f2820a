"""GPU-box diagnostic: HIP backward (gradient) kernel vs torch.autograd oracle."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle import kalman_torch as kt  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

eng = KalmanEngine()
cases = [
    ("c1", W.nile_local_level(B=3), "standard"),
    ("c2", W.arma11(B=6, n=40), "standard"),
    ("headline", W.structural(B=5, n=30), "standard"),
    ("headline_chol", W.structural(B=5, n=30), "cholesky"),
    ("headline_single", W.structural(B=3, n=20), "single"),
    ("c4_m10", W.varmax(B=3, n=15, order=(2, 0)), "standard"),
    ("c4_m10_uni", W.varmax(B=3, n=15, order=(2, 0)), "univariate"),
    ("c4_m15", W.varmax(B=2, n=10, order=(2, 1)), "standard"),
    ("dense_p3m4_fullH", W.random_dense(4, 20, 3, 4, 2, seed=5, h_diag=False), "standard"),
    ("dense_p2m6", W.random_dense(3, 15, 2, 6, 3, seed=6), "standard"),
    ("dense_p9m12", W.random_dense(2, 10, 9, 12, 3, seed=5), "standard"),
]
rng = np.random.default_rng(0)
for name, w, ft in cases:
    B, n, m, p = w["B"], w["n"], w["m"], w["p"]
    y = w["y"].copy()
    if n > 12:
        y[2] = np.nan
        y[5, 0] = np.nan
        y[-1] = np.nan
    g_logp = rng.standard_normal(B)
    g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, p))
    args = (y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    ft_oracle = "standard" if ft in ("cholesky",) else ft
    _, _, gt = kt.logp_and_grads(*args, c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft_oracle)
    for lanes in (0, 16):
        eng.set_lanes_per_draw(lanes)
        res = eng.forward(*args, c=c, d=d, filter_type=ft)
        g = eng.backward(*args, c=c, d=d, predicted_states=res.predicted_states,
                         predicted_covs=res.predicted_covs, g_logp=g_logp, g_ll_obs=g_ll,
                         filter_type=ft, want_cd=True)
        torch.cuda.synchronize()
        errs = {}
        for k in kt.NAMES:
            a, bb = g[k].cpu().numpy(), gt[k]
            if ft == "univariate" and k in ("P0", "Q", "H"):
                # entrywise partials of symmetric inputs differ between the two formulations;
                # symmetric parts agree (tests/test_oracle_torch.py)
                a, bb = a + np.swapaxes(a, -1, -2), bb + np.swapaxes(bb, -1, -2)
            errs[k] = np.max(np.abs(a - bb)) / (1.0 + np.max(np.abs(bb)))
        print(f"{name:16s} {ft:10s} G={lanes:<2d} " + " ".join(f"{k}:{v:.0e}" for k, v in errs.items()),
              flush=True)
eng.set_lanes_per_draw(0)
w = W.structural(B=4096, n=1000)
keys = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")
d = [eng.to_device(w[k]) for k in keys]
res = eng.forward(*d, store_filtered=False, store_ll_obs=False)
gl = torch.ones(4096, device="cuda", dtype=torch.float64)
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    g = eng.backward(*d, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs,
                     g_logp=gl)
    torch.cuda.synchronize()
    print(f"backward headline B=4096 n=1000 m=10: {(time.perf_counter() - t0) * 1e3:.2f} ms")
