import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
n = 300
for m in (4, 6, 8, 10):
    for B in (4096, 32768):
        w = W.random_dense(64, n, 1, m, max(1, m // 2), seed=7)
        rep = B // 64
        dev = [eng.to_device(w["y"])] + [eng.to_device(np.tile(w[k], (rep,) + (1,) * (w[k].ndim - 1))) for k in ("a0", "P0", "T", "Z", "R", "H", "Q")]
        res = eng.forward(*dev, store_filtered=False, store_ll_obs=False)
        ones = torch.ones(B, device="cuda", dtype=torch.float64)
        row = []
        for kern in ("row", "mfma"):
            eng.set_forward_kernel(kern)
            fn = lambda: eng.backward(*dev, predicted_states=res.predicted_states, predicted_covs=res.predicted_covs, g_logp=ones)
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
            row.append(f"{kern} {min(ts)*1e3*1000/n:7.2f}")
        eng.set_forward_kernel("auto")
        print(f"bwd m={m} B={B} (ms per 1000 steps): " + " | ".join(row), flush=True)
        del dev, res
        torch.cuda.empty_cache()
