"""Smoother / gradient timing for 16 < m <= 48 (which kernels serve that range, and how fast)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import workloads as W
from pymc_statespace_amd import KalmanEngine

eng = KalmanEngine()
B, n = int(os.environ.get("NB_B", 4096)), int(os.environ.get("NB_N", 200))
for m in [int(x) for x in os.environ.get("NB_M", "16,20,24,32,40,48").split(",")]:
    w = W.random_dense(64, n, 1, m, max(1, m // 3), seed=7)
    rep = B // 64
    dev = [eng.to_device(w["y"])] + [eng.to_device(np.tile(w[k], (rep,) + (1,) * (w[k].ndim - 1))) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
    res = eng.forward(*dev); torch.cuda.synchronize()
    def tm(f, k=2):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(k): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / k * 1e3 * 1000 / n
    tf = tm(lambda: eng.forward(*dev))
    ts = tm(lambda: eng.smooth(dev[3], dev[5], dev[7], res.filtered_states, res.filtered_covs))
    g = torch.ones(B, dtype=torch.float64, device="cuda")
    tb = tm(lambda: eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=g))
    print(f"m={m} B={B}: per 1000 steps  forward {tf:8.1f} ms   smoother {ts:8.1f} ms   gradient {tb:8.1f} ms", flush=True)
