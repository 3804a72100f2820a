"""backward kernels, p = 1: b4 adjoint (KF_BACKWARD_B4=1) vs the round-1 choice (=0); run once per setting."""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
row = f"KF_BACKWARD_B4={os.environ.get('KF_BACKWARD_B4')}:"
for B in (4096, 512):
    for m in (2, 4, 5, 6, 7, 8, 10, 12, 14):
        w = W.random_dense(B, 1000, 1, m, min(m, 3), seed=m)
        dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
        res = eng.forward(*dev, store_ll_obs=False, store_filtered=False)
        ones = torch.ones(B, device="cuda", dtype=torch.float64)
        tb = timeit(lambda: eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=ones))
        row += f"  B{B} m{m}: {tb:.2f}"
        del dev, res
print(row)
