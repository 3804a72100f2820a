import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
kw = dict(mvn_const="exact", store_ll_obs=True, store_filtered=False, store_predicted=False)
for (B, m, n) in ((4096, 64, 1000), (4096, 56, 1000), (4096, 49, 1000), (4096, 48, 1000), (4096, 40, 1000)):
    w = W.random_dense(B, n, 1, m, 3, seed=m)
    dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    r = eng.forward(*dev, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = eng.forward(*dev, **kw); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"B={B} m={m} n={n}: {dt * 1e3:.1f} ms  ({4.0 * m**3 * B * n / dt / 1e12:.1f} TF)  kernel {eng.forward_kernel_name(B, n, 1, m, 3)}", flush=True)
    del dev, r; torch.cuda.empty_cache()
