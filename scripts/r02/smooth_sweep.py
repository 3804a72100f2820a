"""smoother kernels over m: KF_SMOOTHER_B4 unset (Newton-Schulz block kernel) vs =0 (elimination tile kernel)."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
row = f"KF_SMOOTHER_B4={os.environ.get('KF_SMOOTHER_B4')}:"
for B in (4096, 512):
    for m in (5, 6, 7, 8, 10, 11, 12, 14, 15):
        w = W.random_dense(B, 1000, 1, m, min(m, 3), seed=m)
        dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
        res = eng.forward(*dev, store_ll_obs=False, store_predicted=False)
        t = timeit(lambda: eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs))
        row += f"  B{B} m{m}: {t:.2f}"
        del dev, res
w = W.WORKLOADS["c4"](B=2048)
dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
res = eng.forward(*dev, store_ll_obs=False, store_predicted=False, h_diagonal=True)
t = timeit(lambda: eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs))
row += f"  c4 shard (m=15, B=2048): {t:.2f}"
print(row)
