import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
def timeit(fn, n=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for B in (4096, 8192, 16384):
    for m in (12, 13, 14, 15):
        w = W.random_dense(B, 500, 1, m, min(m, 3), seed=m)
        dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
        row = f"random_dense B={B} m={m} T=500:"
        for ex in (False, True):
            res = eng.forward(*dev, store_ll_obs=False, store_predicted=False, exact_recursion=ex)
            for env in ("0", "1"):
                os.environ["KF_SMOOTHER_B4"] = env
                t = timeit(lambda: eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs))
                row += f"  fwd_exact={ex!s:5s} b4={env}: {t:6.2f}"
            del res
        print(row, flush=True)
        del dev
os.environ.pop("KF_SMOOTHER_B4")
