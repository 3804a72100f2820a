"""kf_forward_wgl.hip (64 < m <= 96, P resident in LDS) against the numpy oracle and the HBM-workspace path (KF_FORWARD_WGL=0); timing."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import workloads as W
from oracle import kalman_numpy as kn
from pymc_statespace_amd import KalmanEngine
KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()
kw = dict(mvn_const="exact", store_ll_obs=True, store_filtered=False, store_predicted=False)
for m in (65, 70, 79, 80, 81, 90, 96):
    w = W.random_dense(5, 120, 1, m, 4, seed=300 + m)
    y = w["y"].copy(); y[[0, 17, 60]] = np.nan
    rng = np.random.default_rng(m)
    c = 0.1 * rng.standard_normal((5, m)); d = 0.1 * rng.standard_normal((5, 1))
    args = (y,) + tuple(w[k] for k in KEYS)
    ref = kn.kalman_filter(*args, c=c, d=d, mvn_const="exact")
    os.environ["KF_FORWARD_WGL"] = "1"
    res = eng.forward(*args, c=c, d=d, **kw)
    os.environ["KF_FORWARD_WGL"] = "0"
    big = eng.forward(*args, c=c, d=d, **kw)
    e1 = np.abs(res.logp.cpu().numpy() - ref["logp"]).max() / np.abs(ref["logp"]).max()
    e2 = np.abs(res.ll_obs.cpu().numpy() - ref["ll_obs"]).max()
    e3 = (res.logp - big.logp).abs().max().item() / np.abs(ref["logp"]).max()
    print(f"m={m}: rel dlogp vs oracle {e1:.1e}, ll_obs abs {e2:.1e}, vs workspace path {e3:.1e}, info {res.info.cpu().numpy().any()}", flush=True)
os.environ["KF_FORWARD_WGL"] = "1"
for (B, m, n) in ((4096, 80, 1000), (4096, 96, 1000), (4096, 70, 1000), (512, 80, 1000)):
    w = W.random_dense(B, n, 1, m, 3, seed=m)
    dev = [eng.to_device(w[k]) for k in ("y",) + KEYS]
    for flag in ("1", "0"):
        os.environ["KF_FORWARD_WGL"] = flag
        if flag == "0" and B * m > 4096 * 80: continue
        r = eng.forward(*dev, **kw); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = eng.forward(*dev, **kw); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"B={B} m={m} n={n} KF_FORWARD_WGL={flag}: {dt * 1e3:.1f} ms  ({4.0 * m**3 * B * n / dt / 1e12:.1f} TF issued)", flush=True)
    del dev, r
    torch.cuda.empty_cache()
