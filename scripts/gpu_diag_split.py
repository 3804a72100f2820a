"""GPU-box diagnostic: row-split forward kernels (all compiled (M,H)) vs oracle + headline timing."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from oracle import kalman_numpy as kn  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

COMBOS = [(10, 1), (10, 2), (10, 3), (10, 5), (8, 1), (8, 2), (8, 4), (6, 1), (6, 3), (4, 1), (4, 2), (2, 1)]


def relerr(a, b):
    return float(np.max(np.abs(a - b) / (1.0 + np.abs(b))))


eng = KalmanEngine()
eng.set_forward_kernel("row")
for m, H in COMBOS:
    B = 41
    w = W.random_dense(B, 40, 1, m, max(1, m // 2), seed=100 + m)
    y = w["y"].copy()
    y[[3, 17]] = np.nan
    args = (y, w["a0"], w["P0"], w["T"], w["Z"], w["R"], w["H"], w["Q"])
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((B, m))
    d = 0.1 * rng.standard_normal((B, 1))
    eng.set_lanes_per_draw(m * H)
    for ft in ("standard", "univariate", "single"):
        res = eng.forward(*args, c=c, d=d, filter_type=ft, mvn_const="exact")
        o = kn.kalman_filter(*args, c=c, d=d, filter_type=ft, mvn_const="exact")
        print(f"m={m:2d} H={H} {ft:10s} |dlogp| {np.max(np.abs(res.logp.cpu().numpy() - o['logp'])):.1e} "
              f"ll {relerr(res.ll_obs.cpu().numpy(), o['ll_obs']):.1e} "
              f"fa {relerr(res.filtered_states.cpu().numpy(), o['filtered_states']):.1e} "
              f"pa {relerr(res.predicted_states.cpu().numpy(), o['predicted_states']):.1e} "
              f"fP {relerr(res.filtered_covs.cpu().numpy(), o['filtered_covs']):.1e} "
              f"pP {relerr(res.predicted_covs.cpu().numpy(), o['predicted_covs']):.1e} "
              f"info {int(res.info.abs().sum())}", flush=True)
keys = ("y", "a0", "P0", "T", "Z", "R", "H", "Q")
for name, w in (("headline", W.structural(B=4096, n=1000)), ("c3", W.structural(B=8192, n=2000))):
    d = [eng.to_device(w[k]) for k in keys]
    for H in (1, 2, 3, 5):
        eng.set_lanes_per_draw(10 * H)
        for full in (False, True):
            kw = dict(store_ll_obs=full, store_filtered=full, store_predicted=full)
            res = eng.forward(*d, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                eng.forward(*d, out=res, **kw)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3
            print(f"{name:9s} B={w['B']} n={w['n']} m={w['m']} H={H} full_tape={full}: "
                  f"{dt * 1e3:.3f} ms -> {w['B'] / dt:.3e} evals/s", flush=True)
        del res
    del d
    torch.cuda.empty_cache()
