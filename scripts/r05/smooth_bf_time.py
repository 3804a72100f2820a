"""kf_smoother_from_filter (kf_smoother_bf.hip) against kf_smoother at the BASELINE shapes: time and largest deviation."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import workloads as W  # noqa: E402
from pymc_statespace_amd import KalmanEngine  # noqa: E402

KEYS = ("a0", "P0", "T", "Z", "R", "H", "Q")
eng = KalmanEngine()


def ms(fn, reps=5):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


cases = [("headline", lambda: W.structural(B=4096, n=1000), "standard"), ("headline/8", lambda: W.structural(B=512, n=1000), "standard"),
         ("c3", lambda: W.structural(B=8192, n=2000), "cholesky"), ("c2", lambda: W.arma11(B=4096, n=1000), "standard"),
         ("m5", lambda: W.random_dense(4096, 1000, 1, 5, 3, seed=3), "standard"), ("m14", lambda: W.random_dense(4096, 500, 1, 14, 6, seed=3), "standard")]
if len(sys.argv) > 1:
    cases = [c for c in cases if c[0] in sys.argv[1:]]
for name, mk, ft in cases:
    w = mk()
    d = [eng.to_device(w["y"])] + [eng.to_device(w[k]) for k in KEYS]
    res = eng.forward(*d, filter_type=ft, mvn_const="exact", store_ll_obs=False)
    t_bf = ms(lambda: eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs))
    sa, sP = eng.smooth_from_filter(d[0], d[3], d[4], d[6], res.predicted_states, res.predicted_covs)
    t_rts = ms(lambda: eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs))
    xa, xP = eng.smooth(d[3], d[5], d[7], res.filtered_states, res.filtered_covs)
    torch.cuda.synchronize()
    B = sa.shape[0]
    ea = float(((sa - xa).abs().reshape(B, -1).max(dim=1).values / (1e-300 + xa.abs().reshape(B, -1).max(dim=1).values)).max())
    eP = float(((sP - xP).abs().reshape(B, -1).max(dim=1).values / (1e-300 + xP.abs().reshape(B, -1).max(dim=1).values)).max())
    gb = 2 * sa.numel() * 8 / 1e9 + 2 * sP.numel() * 8 / 1e9
    print(f"{name:11s} from-filter {t_bf:7.3f} ms ({gb / t_bf / 8:.3f} of 8 TB/s)   kf_smoother {t_rts:7.3f} ms   "
          f"largest deviation (rel. to the draw's largest entry): states {ea:.2e}  covariances {eP:.2e}", flush=True)
    del res, sa, sP, xa, xP
