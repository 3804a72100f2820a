"""GPU-box diagnostic: register-resident backward kernels (p=1, even m<=10) vs torch.autograd."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as W
from oracle import kalman_torch as kt
from pymc_statespace_amd import KalmanEngine

COMBOS = [(10, 1), (10, 2), (10, 3), (10, 5), (8, 1), (8, 2), (8, 4), (6, 1), (6, 3), (4, 1), (4, 2), (2, 1)]
eng = KalmanEngine()
rng = np.random.default_rng(0)
for m, H in COMBOS:
    B, n = 13, 25
    w = W.random_dense(B, n, 1, m, max(1, m // 2), seed=300 + m)
    y = w["y"].copy(); y[2] = np.nan; y[-1] = np.nan
    g_logp = rng.standard_normal(B); g_ll = rng.standard_normal((B, n))
    c = 0.1 * rng.standard_normal((B, m)); d = 0.1 * rng.standard_normal((B, 1))
    args = (y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    for ft in ("standard", "single"):
        _, _, gt = kt.logp_and_grads(*args, c=c, d=d, g_logp=g_logp, g_ll_obs=g_ll, filter_type=ft)
        eng.set_lanes_per_draw(m * H)
        res = eng.forward(*args, c=c, d=d, filter_type=ft)
        g = eng.backward(*args, c=c, d=d, predicted_states=res.predicted_states,
                         predicted_covs=res.predicted_covs, g_logp=g_logp, g_ll_obs=g_ll,
                         filter_type=ft, want_cd=True)
        torch.cuda.synchronize()
        errs = {k: np.max(np.abs(g[k].cpu().numpy() - gt[k])) / (1.0 + np.max(np.abs(gt[k]))) for k in kt.NAMES}
        print(f"m={m:2d} H={H} {ft:8s} " + " ".join(f"{k}:{v:.0e}" for k, v in errs.items()), flush=True)
w = W.structural(B=4096, n=1000)
d = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
eng.set_lanes_per_draw(0)
res = eng.forward(*d, store_filtered=False, store_ll_obs=False)
gl = torch.ones(4096, device="cuda", dtype=torch.float64)
for H in (1, 2, 3, 5):
    eng.set_lanes_per_draw(10 * H)
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        g = eng.backward(*d, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=gl)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"backward headline B=4096 n=1000 m=10 H={H}: {min(ts)*1e3:.2f} ms")
