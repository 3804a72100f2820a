"""64 < m <= 128 (the HBM-workspace path): forward, logp only, 4096 draws x 1000 steps at m = 80, and m = 96 / 128 at 1024 draws."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from pymc_statespace_amd import KalmanEngine
eng = KalmanEngine()
for (B, m, n) in ((4096, 80, 1000), (1024, 96, 1000), (1024, 128, 1000)):
    w = W.random_dense(B, n, 1, m, 3, seed=m)
    dev = [eng.to_device(w[k]) for k in ("y", "a0", "P0", "T", "Z", "R", "H", "Q")]
    kw = dict(mvn_const="exact", store_ll_obs=True, store_filtered=False, store_predicted=False)
    for fuse in ("1", "0"):
        os.environ["KF_BIG_FUSE"] = fuse
        r = eng.forward(*dev, **kw); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = eng.forward(*dev, **kw); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B} m={m} n={n} KF_BIG_FUSE={fuse}: {dt * 1e3:.1f} ms  ({4.0 * m**3 * B * n / dt / 1e12:.1f} TF issued)  kernel {eng.forward_kernel_name(B, n, 1, m, 3)}", flush=True)
    del dev, r
    torch.cuda.empty_cache()
