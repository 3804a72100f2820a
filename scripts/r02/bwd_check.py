"""b4 adjoint (kf_backward_b4.hip) against torch autograd + timing at the headline shape."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_torch as kt
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
worst = 0.0
for m in (1, 2, 3, 5, 6, 7, 9, 10, 11, 13, 14):
    for ft in ("standard", "univariate", "single"):
        B, n, r = 3, 70, max(1, m // 2)
        w = W.random_dense(B, n, 1, m, r, seed=50 + m)
        rng = np.random.default_rng(m)
        y = w["y"].copy(); y[[0, 5, 64, 69]] = np.nan
        c = 0.1 * rng.standard_normal((B, m)); d = 0.1 * rng.standard_normal((B, 1))
        gl = rng.standard_normal(B); gll = 0.3 * rng.standard_normal((B, n))
        args = (y,) + tuple(w[k] for k in KEYS) + (c, d)
        _, _, gt = kt.logp_and_grads(*args, filter_type=ft, g_logp=gl, g_ll_obs=gll)
        res = eng.forward(*args, filter_type=ft)
        g = eng.backward(*args, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=gl,
                         g_ll_obs=gll, filter_type=ft, want_cd=True)
        row = f"m={m:2d} {ft:10s}"
        for k in KEYS + ("c", "d"):
            a_, b_ = g[k].cpu().numpy(), gt[k]
            if ft == "univariate" and k in ("P0", "Q", "H"):
                a_, b_ = a_ + np.swapaxes(a_, -1, -2), b_ + np.swapaxes(b_, -1, -2)
            e = np.max(np.abs(a_ - b_)) / (1 + np.max(np.abs(b_)))
            worst = max(worst, e if np.isfinite(e) else 1e9)
            row += f" {k}:{e:.1e}"
        print(row, flush=True)
print("worst", worst)
w = W.WORKLOADS["headline"]()
dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
res = eng.forward(*dev, store_ll_obs=False, store_filtered=False)
ones = torch.ones(w["B"], device="cuda", dtype=torch.float64)
def timeit(fn, n=6):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
tb = timeit(lambda: eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=ones))
tf = timeit(lambda: eng.forward(*dev, store_ll_obs=False, store_filtered=False, out=res))
print(f"headline: forward (predicted tapes only) {tf:.3f} ms  backward {tb:.3f} ms  KF_BACKWARD_B4={os.environ.get('KF_BACKWARD_B4')}")
