import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
n = 100
for m in (16, 17, 20, 24, 32, 40):
    B = 4096
    w = W.random_dense(64, n, 1, m, m // 2, seed=7)
    rep = B // 64
    dev = [eng.to_device(w["y"])] + [eng.to_device(np.tile(w[k], (rep,) + (1,) * (w[k].ndim - 1))) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    kw = dict(store_ll_obs=False, store_filtered=False, store_predicted=False)
    out = eng.forward(*dev, **kw); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); out = eng.forward(*dev, out=out, **kw); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"m={m} B={B}: {eng.forward_kernel_name(B, n, 1, m, m // 2)} logp-only {min(ts)*1e3*1000/n:.1f} ms per 1000 steps", flush=True)
