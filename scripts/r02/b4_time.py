import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
eng.set_forward_kernel(os.environ.get("KERN", "b4"))
w = W.WORKLOADS["headline"]()
dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
full = eng.forward(*dev, mvn_const="exact")
lo = eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False)
pr = eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=True)
print(os.environ.get("KF_LIB_PATH", "default")[-22:], "full %.3f  logp %.3f  pred-only %.3f ms" % (
    timeit(lambda: eng.forward(*dev, mvn_const="exact", out=full)),
    timeit(lambda: eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=False, out=lo)),
    timeit(lambda: eng.forward(*dev, mvn_const="exact", store_ll_obs=False, store_filtered=False, store_predicted=True, out=pr))))
