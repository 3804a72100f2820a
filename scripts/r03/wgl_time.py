"""Per-step time of kf_forward_wgl.hip: one round of 256 draws (one workgroup per CU), 500 steps."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
kw = dict(mvn_const="exact", store_ll_obs=True, store_filtered=False, store_predicted=False)
for m in (80, 70, 96, 88):
    B, n = 256, 500
    w = W.random_dense(B, n, 1, m, 3, seed=m)
    dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    r = eng.forward(*dev, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = eng.forward(*dev, **kw); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"m={m}: {dt / n * 1e6:.1f} us per step", flush=True)
