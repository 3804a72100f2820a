import sys, time, os
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
from pymc_statespace_amd.engine import FilterResult
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine(); eng.set_forward_kernel("b4")
w = W.WORKLOADS["headline"]()
dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
B, n, m = 4096, 1000, 10
d = dev[0].device
out = FilterResult(logp=torch.empty(B, device=d, dtype=torch.float64), ll_obs=torch.empty(B, n, device=d, dtype=torch.float64),
    filtered_states=torch.empty(B, n, m, device=d, dtype=torch.float64), predicted_states=torch.empty(B, n + 1, m, device=d, dtype=torch.float64),
    filtered_covs=torch.empty(B, n + 2, 128, device=d, dtype=torch.float64), predicted_covs=torch.empty(B, n + 2, 128, device=d, dtype=torch.float64),
    info=torch.empty(B, device=d, dtype=torch.int32))
def timeit(fn, k=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
print(os.environ.get("KF_LIB_PATH", "default")[-40:], "full (padded tapes) %.3f ms" % timeit(lambda: eng.forward(*dev, mvn_const="exact", out=out)))
